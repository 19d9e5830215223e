/*
 * gsr_oracle.c -- CPU restatement of the Gaussian-splatting rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (gaussian-splatting-toolkit_amd/) never links or imports it.
 *
 * What it restates (paths relative to
 * /root/reference/gs_toolkit/gs_components/rasterizer/):
 *   cuda/csrc/forward.cu   project (13-90), map_intersects (94-127),
 *                          bin edges (132-154), rasterize_forward (278-395),
 *                          nd_rasterize_forward (159-276), EWA (398-442),
 *                          scale_rot_to_cov3d (445-464)
 *   cuda/csrc/backward.cu  rasterize_backward (133-303), nd (23-131),
 *                          project backward (305-347), EWA vjp (350-423),
 *                          scale/rot vjp (427-453)
 *   cuda/csrc/helpers.cuh  ndc2pix/bbox/bounds/vjps/quat (7-219)
 *   cuda/csrc/sh.cuh       SH colour + vjp (33-224)
 *   utils.py               cumsum (106-125), sort+gather (128-182)
 *
 * Pinning: the reference ships no tests and no golden vectors for this path
 * ("parity unpinned" by the reference itself).  This restatement is pinned
 * against outputs of the reference's own pure-PyTorch implementation
 * (_torch_impl.py) generated in the build container and committed under
 * tests/golden/ (see tests/golden/make_golden.py).  The backward has no
 * reference CPU implementation; it is pinned against torch.autograd through
 * _torch_impl.py (same fixtures) and by finite differences.
 *
 * Deliberate deviations from the CUDA source, all documented in DESIGN.md:
 *   - gradient sums are accumulated in double and rounded once (the CUDA
 *     kernel uses float atomics in nondeterministic order);
 *   - float->int conversions saturate (GPU semantics) instead of being UB;
 *   - the sort is stable (ties keep emission order); torch.sort is not;
 *   - the N-channel path accumulates in fp32 (the CUDA path uses __half).
 *
 * Plain C99 + optional OpenMP.  Matrices are row-major as in the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define GSR_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ utils */

GSR_API int gsr_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

GSR_API void gsr_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* float -> int32 truncation with GPU (saturating, NaN->0) semantics */
static inline int f2i_sat(float v) {
  if (v != v) return 0;
  if (v >= 2147483520.f) return 2147483647;
  if (v <= -2147483648.f) return (-2147483647 - 1);
  return (int)v;
}
static inline int clampi(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline float fminf_(float a, float b) { return a < b ? a : b; }

/* -------------------------------------------------------- small 3x3 math */

typedef struct { float m[3][3]; } m3; /* row-major: m[row][col] */

static inline m3 m3_mul(const m3 *A, const m3 *B) {
  m3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      C.m[i][j] = A->m[i][0] * B->m[0][j] + A->m[i][1] * B->m[1][j] +
                  A->m[i][2] * B->m[2][j];
  return C;
}
static inline m3 m3_T(const m3 *A) {
  m3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A->m[j][i];
  return C;
}

/* helpers.cuh:144-159 -- quaternion (w,x,y,z) -> rotation, renormalised */
static inline m3 quat_to_R(const float *q) {
  float s = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
  m3 R;
  R.m[0][0] = 1.f - 2.f * (y * y + z * z);
  R.m[0][1] = 2.f * (x * y - w * z);
  R.m[0][2] = 2.f * (x * z + w * y);
  R.m[1][0] = 2.f * (x * y + w * z);
  R.m[1][1] = 1.f - 2.f * (x * x + z * z);
  R.m[1][2] = 2.f * (y * z - w * x);
  R.m[2][0] = 2.f * (x * z - w * y);
  R.m[2][1] = 2.f * (y * z + w * x);
  R.m[2][2] = 1.f - 2.f * (x * x + y * y);
  return R;
}

/* forward.cu:445-464 */
static inline void cov3d_from_scale_rot(const float *scale, float glob,
                                        const float *quat, float *cov3d) {
  m3 R = quat_to_R(quat);
  m3 M;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M.m[i][j] = R.m[i][j] * (glob * scale[j]);
  m3 Mt = m3_T(&M);
  m3 S = m3_mul(&M, &Mt);
  cov3d[0] = S.m[0][0];
  cov3d[1] = S.m[0][1];
  cov3d[2] = S.m[0][2];
  cov3d[3] = S.m[1][1];
  cov3d[4] = S.m[1][2];
  cov3d[5] = S.m[2][2];
}

/* helpers.cuh:23-34 + 11-21 : tile bbox, inclusive min / exclusive max */
static inline void tile_bbox(float cx, float cy, float radius, int tiles_x,
                             int tiles_y, int bw, int *minx, int *miny,
                             int *maxx, int *maxy) {
  float tcx = cx / (float)bw, tcy = cy / (float)bw;
  float tr = radius / (float)bw;
  *minx = clampi(f2i_sat(tcx - tr), 0, tiles_x);
  *maxx = clampi(f2i_sat(tcx + tr + 1.f), 0, tiles_x);
  *miny = clampi(f2i_sat(tcy - tr), 0, tiles_y);
  *maxy = clampi(f2i_sat(tcy + tr + 1.f), 0, tiles_y);
}

/* helpers.cuh:36-59 */
static inline int cov2d_bounds(const float *cov2d, float *conic,
                               float *radius) {
  float det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1];
  if (det == 0.f) return 0;
  float inv_det = 1.f / det;
  conic[0] = cov2d[2] * inv_det;
  conic[1] = -cov2d[1] * inv_det;
  conic[2] = cov2d[0] * inv_det;
  float b = 0.5f * (cov2d[0] + cov2d[2]);
  float disc = sqrtf(fmaxf_(0.1f, b * b - det));
  float v1 = b + disc, v2 = b - disc;
  *radius = ceilf(3.f * sqrtf(fmaxf_(v1, v2)));
  return 1;
}

/* ------------------------------------------------------- project forward */

/* forward.cu:13-90.  viewmat: >=12 floats (3x4 rows), projmat: 16 floats.
 * Every output element of every Gaussian is written (zeros where the CUDA
 * kernel returns early on a torch::zeros buffer, bindings.cu:126-139). */
GSR_API void gsr_oracle_project_forward(
    int n, const float *means3d, const float *scales, float glob_scale,
    const float *quats, const float *viewmat, const float *projmat, float fx,
    float fy, float cx, float cy, int img_h, int img_w, int bw,
    float clip_thresh, float *cov3d, float *xys, float *depths, int *radii,
    float *conics, float *compensation, int *num_tiles_hit) {
  const int tiles_x = (img_w + bw - 1) / bw, tiles_y = (img_h + bw - 1) / bw;
  const float tan_fovx = 0.5f * (float)img_w / fx;
  const float tan_fovy = 0.5f * (float)img_h / fy;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    float *c3 = cov3d + 6 * i;
    for (int k = 0; k < 6; ++k) c3[k] = 0.f;
    xys[2 * i] = xys[2 * i + 1] = 0.f;
    depths[i] = 0.f;
    radii[i] = 0;
    conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0.f;
    compensation[i] = 0.f;
    num_tiles_hit[i] = 0;

    const float *p = means3d + 3 * i;
    /* helpers.cuh:212-219 near-plane cull (<=) */
    float tx = viewmat[0] * p[0] + viewmat[1] * p[1] + viewmat[2] * p[2] + viewmat[3];
    float ty = viewmat[4] * p[0] + viewmat[5] * p[1] + viewmat[6] * p[2] + viewmat[7];
    float tz = viewmat[8] * p[0] + viewmat[9] * p[1] + viewmat[10] * p[2] + viewmat[11];
    if (tz <= clip_thresh) continue;

    cov3d_from_scale_rot(scales + 3 * i, glob_scale, quats + 4 * i, c3);

    /* forward.cu:398-442 EWA projection */
    float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    float ex = tz * fminf_(limx, fmaxf_(-limx, tx / tz));
    float ey = tz * fminf_(limy, fmaxf_(-limy, ty / tz));
    float rz = 1.f / tz, rz2 = rz * rz;
    m3 J = {{{fx * rz, 0.f, -fx * ex * rz2}, {0.f, fy * rz, -fy * ey * rz2}, {0.f, 0.f, 0.f}}};
    m3 W = {{{viewmat[0], viewmat[1], viewmat[2]},
             {viewmat[4], viewmat[5], viewmat[6]},
             {viewmat[8], viewmat[9], viewmat[10]}}};
    m3 T = m3_mul(&J, &W);
    m3 V = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
    m3 TV = m3_mul(&T, &V);
    m3 Tt = m3_T(&T);
    m3 C = m3_mul(&TV, &Tt);
    float c00 = C.m[0][0], c11 = C.m[1][1], c01 = C.m[0][1];
    float det_orig = c00 * c11 - c01 * c01;
    float cov2d[3] = {c00 + 0.3f, c01, c11 + 0.3f};
    float det_blur = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1];
    float comp = sqrtf(fmaxf_(0.f, det_orig / det_blur));

    float conic[3], radius;
    if (!cov2d_bounds(cov2d, conic, &radius)) continue;
    conics[3 * i] = conic[0];
    conics[3 * i + 1] = conic[1];
    conics[3 * i + 2] = conic[2];

    /* helpers.cuh:114-122 */
    float hx = projmat[0] * p[0] + projmat[1] * p[1] + projmat[2] * p[2] + projmat[3];
    float hy = projmat[4] * p[0] + projmat[5] * p[1] + projmat[6] * p[2] + projmat[7];
    float hw = projmat[12] * p[0] + projmat[13] * p[1] + projmat[14] * p[2] + projmat[15];
    float rw = 1.f / (hw + 1e-6f);
    float px = 0.5f * (float)img_w * (hx * rw) + cx - 0.5f;
    float py = 0.5f * (float)img_h * (hy * rw) + cy - 0.5f;

    int minx, miny, maxx, maxy;
    tile_bbox(px, py, radius, tiles_x, tiles_y, bw, &minx, &miny, &maxx, &maxy);
    int area = (maxx - minx) * (maxy - miny);
    if (area <= 0) continue;

    num_tiles_hit[i] = area;
    depths[i] = tz;
    radii[i] = f2i_sat(radius);
    xys[2 * i] = px;
    xys[2 * i + 1] = py;
    compensation[i] = comp;
  }
}

/* ------------------------------------------------------ project backward */

/* backward.cu:305-453 + helpers.cuh:62-90,125-142,161-200.
 * Outputs are fully written (zero where radii<=0, bindings.cu:182-191). */
GSR_API void gsr_oracle_project_backward(
    int n, const float *means3d, const float *scales, float glob_scale,
    const float *quats, const float *viewmat, const float *projmat, float fx,
    float fy, float cx, float cy, int img_h, int img_w, const float *cov3d,
    const int *radii, const float *conics, const float *compensation,
    const float *v_xy, const float *v_depth, const float *v_conic,
    const float *v_compensation, float *v_cov2d, float *v_cov3d,
    float *v_mean3d, float *v_scale, float *v_quat) {
  (void)cx;
  (void)cy;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    float *o2 = v_cov2d + 3 * i, *o3 = v_cov3d + 6 * i, *om = v_mean3d + 3 * i;
    float *os = v_scale + 3 * i, *oq = v_quat + 4 * i;
    o2[0] = o2[1] = o2[2] = 0.f;
    for (int k = 0; k < 6; ++k) o3[k] = 0.f;
    om[0] = om[1] = om[2] = 0.f;
    os[0] = os[1] = os[2] = 0.f;
    oq[0] = oq[1] = oq[2] = oq[3] = 0.f;
    if (radii[i] <= 0) continue;
    const float *p = means3d + 3 * i;

    /* project_pix_vjp */
    float hx = projmat[0] * p[0] + projmat[1] * p[1] + projmat[2] * p[2] + projmat[3];
    float hy = projmat[4] * p[0] + projmat[5] * p[1] + projmat[6] * p[2] + projmat[7];
    float hw = projmat[12] * p[0] + projmat[13] * p[1] + projmat[14] * p[2] + projmat[15];
    float rw = 1.f / (hw + 1e-6f);
    float vnx = 0.5f * (float)img_w * v_xy[2 * i];
    float vny = 0.5f * (float)img_h * v_xy[2 * i + 1];
    float vt0 = vnx * rw, vt1 = vny * rw;
    float vt3 = -(vnx * hx + vny * hy) * rw * rw;
    float vm[3];
    for (int k = 0; k < 3; ++k)
      vm[k] = projmat[k] * vt0 + projmat[4 + k] * vt1 + projmat[12 + k] * vt3;

    /* depth = row 2 of viewmat . p */
    float vz = v_depth[i];
    vm[0] += viewmat[8] * vz;
    vm[1] += viewmat[9] * vz;
    vm[2] += viewmat[10] * vz;

    /* cov2d_to_conic_vjp : v_cov2d = -X G X */
    const float *co = conics + 3 * i;
    const float *vc = v_conic + 3 * i;
    float X00 = co[0], X01 = co[1], X11 = co[2];
    float G00 = vc[0], G01 = 0.5f * vc[1], G11 = vc[2];
    /* XG */
    float A00 = X00 * G00 + X01 * G01, A01 = X00 * G01 + X01 * G11;
    float A10 = X01 * G00 + X11 * G01, A11 = X01 * G01 + X11 * G11;
    /* -(XG)X */
    float S00 = -(A00 * X00 + A01 * X01), S01 = -(A00 * X01 + A01 * X11);
    float S10 = -(A10 * X00 + A11 * X01), S11 = -(A10 * X01 + A11 * X11);
    float vcov2d[3] = {S00, S01 + S10, S11};

    /* cov2d_to_compensation_vjp */
    {
      float comp = compensation[i], vcomp = v_compensation[i];
      float inv_det = co[0] * co[2] - co[1] * co[1];
      float om2 = 1.f - comp * comp;
      float vsq = vcomp * 0.5f / (comp + 1e-6f);
      vcov2d[0] += vsq * (om2 * co[0] - 0.3f * inv_det);
      vcov2d[1] += 2.f * vsq * (om2 * co[1]);
      vcov2d[2] += vsq * (om2 * co[2] - 0.3f * inv_det);
    }
    o2[0] = vcov2d[0];
    o2[1] = vcov2d[1];
    o2[2] = vcov2d[2];

    /* project_cov3d_ewa_vjp (no fov clamp in the backward, backward.cu:367) */
    m3 W = {{{viewmat[0], viewmat[1], viewmat[2]},
             {viewmat[4], viewmat[5], viewmat[6]},
             {viewmat[8], viewmat[9], viewmat[10]}}};
    float tx = W.m[0][0] * p[0] + W.m[0][1] * p[1] + W.m[0][2] * p[2] + viewmat[3];
    float ty = W.m[1][0] * p[0] + W.m[1][1] * p[1] + W.m[1][2] * p[2] + viewmat[7];
    float tz = W.m[2][0] * p[0] + W.m[2][1] * p[1] + W.m[2][2] * p[2] + viewmat[11];
    float rz = 1.f / tz, rz2 = rz * rz, rz3 = rz2 * rz;
    m3 J = {{{fx * rz, 0.f, -fx * tx * rz2}, {0.f, fy * rz, -fy * ty * rz2}, {0.f, 0.f, 0.f}}};
    const float *c3 = cov3d + 6 * i;
    m3 V = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
    m3 Gc = {{{vcov2d[0], 0.5f * vcov2d[1], 0.f}, {0.5f * vcov2d[1], vcov2d[2], 0.f}, {0.f, 0.f, 0.f}}};
    m3 T = m3_mul(&J, &W);
    m3 Tt = m3_T(&T);
    m3 TtG = m3_mul(&Tt, &Gc);
    m3 vV = m3_mul(&TtG, &T);
    o3[0] = vV.m[0][0];
    o3[1] = vV.m[0][1] + vV.m[1][0];
    o3[2] = vV.m[0][2] + vV.m[2][0];
    o3[3] = vV.m[1][1];
    o3[4] = vV.m[1][2] + vV.m[2][1];
    o3[5] = vV.m[2][2];
    /* v_T = G T V^T + G^T T V */
    m3 GT = m3_mul(&Gc, &T);
    m3 Vt = m3_T(&V);
    m3 P1 = m3_mul(&GT, &Vt);
    m3 Gt = m3_T(&Gc);
    m3 GtT = m3_mul(&Gt, &T);
    m3 P2 = m3_mul(&GtT, &V);
    m3 vT;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) vT.m[r][c] = P1.m[r][c] + P2.m[r][c];
    m3 Wt = m3_T(&W);
    m3 vJ = m3_mul(&vT, &Wt);
    float vtx = -fx * rz2 * vJ.m[0][2];
    float vty = -fy * rz2 * vJ.m[1][2];
    float vtz = -fx * rz2 * vJ.m[0][0] + 2.f * fx * tx * rz3 * vJ.m[0][2] -
                fy * rz2 * vJ.m[1][1] + 2.f * fy * ty * rz3 * vJ.m[1][2];
    /* v_mean += W^T v_t */
    vm[0] += vtx * W.m[0][0] + vty * W.m[1][0] + vtz * W.m[2][0];
    vm[1] += vtx * W.m[0][1] + vty * W.m[1][1] + vtz * W.m[2][1];
    vm[2] += vtx * W.m[0][2] + vty * W.m[1][2] + vtz * W.m[2][2];
    om[0] = vm[0];
    om[1] = vm[1];
    om[2] = vm[2];

    /* scale_rot_to_cov3d_vjp */
    m3 vS = {{{o3[0], 0.5f * o3[1], 0.5f * o3[2]},
              {0.5f * o3[1], o3[3], 0.5f * o3[4]},
              {0.5f * o3[2], 0.5f * o3[4], o3[5]}}};
    const float *q = quats + 4 * i;
    const float *sc = scales + 3 * i;
    m3 R = quat_to_R(q);
    m3 M;
    float s3[3] = {glob_scale * sc[0], glob_scale * sc[1], glob_scale * sc[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M.m[r][c] = R.m[r][c] * s3[c];
    m3 vM = m3_mul(&vS, &M);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) vM.m[r][c] *= 2.f;
    for (int c = 0; c < 3; ++c)
      os[c] = (R.m[0][c] * vM.m[0][c] + R.m[1][c] * vM.m[1][c] + R.m[2][c] * vM.m[2][c]) * glob_scale;
    m3 vR;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) vR.m[r][c] = vM.m[r][c] * s3[c];
    /* quat_to_rotmat_vjp: q treated as unit after renormalisation */
    float s = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    oq[0] = 2.f * (x * (vR.m[2][1] - vR.m[1][2]) + y * (vR.m[0][2] - vR.m[2][0]) +
                   z * (vR.m[1][0] - vR.m[0][1]));
    oq[1] = 2.f * (-2.f * x * (vR.m[1][1] + vR.m[2][2]) + y * (vR.m[1][0] + vR.m[0][1]) +
                   z * (vR.m[2][0] + vR.m[0][2]) + w * (vR.m[2][1] - vR.m[1][2]));
    oq[2] = 2.f * (x * (vR.m[1][0] + vR.m[0][1]) - 2.f * y * (vR.m[0][0] + vR.m[2][2]) +
                   z * (vR.m[2][1] + vR.m[1][2]) + w * (vR.m[0][2] - vR.m[2][0]));
    oq[3] = 2.f * (x * (vR.m[2][0] + vR.m[0][2]) + y * (vR.m[2][1] + vR.m[1][2]) -
                   2.f * z * (vR.m[0][0] + vR.m[1][1]) + w * (vR.m[1][0] - vR.m[0][1]));
  }
}

/* -------------------------------------------------- spherical harmonics */

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                               0.31539156525252005f, -1.0925484305920792f,
                               0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                               -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};
static const float SH_C4[9] = {2.5033429417967046f,  -1.7701307697799304f,
                               0.9461746957575601f,  -0.6690465435572892f,
                               0.10578554691520431f, -0.6690465435572892f,
                               0.47308734787878004f, -1.7701307697799304f,
                               0.6258357354491761f};

static inline int sh_bases(int degree) {
  return degree == 0 ? 1 : degree == 1 ? 4 : degree == 2 ? 9 : degree == 3 ? 16 : 25;
}

/* basis values for `deg` (sh.cuh:33-98 factored into a basis vector);
 * returns number of bases filled */
static inline int sh_basis_vec(int deg, const float *dir, float *B) {
  B[0] = SH_C0;
  if (deg < 1) return 1;
  float nrm = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  float x = dir[0] / nrm, y = dir[1] / nrm, z = dir[2] / nrm;
  B[1] = -SH_C1 * y;
  B[2] = SH_C1 * z;
  B[3] = -SH_C1 * x;
  if (deg < 2) return 4;
  float xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
  B[4] = SH_C2[0] * xy;
  B[5] = SH_C2[1] * yz;
  B[6] = SH_C2[2] * (2.f * zz - xx - yy);
  B[7] = SH_C2[3] * xz;
  B[8] = SH_C2[4] * (xx - yy);
  if (deg < 3) return 9;
  B[9] = SH_C3[0] * y * (3.f * xx - yy);
  B[10] = SH_C3[1] * xy * z;
  B[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
  B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
  B[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
  B[14] = SH_C3[5] * z * (xx - yy);
  B[15] = SH_C3[6] * x * (xx - 3.f * yy);
  if (deg < 4) return 16;
  B[16] = SH_C4[0] * xy * (xx - yy);
  B[17] = SH_C4[1] * yz * (3.f * xx - yy);
  B[18] = SH_C4[2] * xy * (7.f * zz - 1.f);
  B[19] = SH_C4[3] * yz * (7.f * zz - 3.f);
  B[20] = SH_C4[4] * (zz * (35.f * zz - 30.f) + 3.f);
  B[21] = SH_C4[5] * xz * (7.f * zz - 3.f);
  B[22] = SH_C4[6] * (xx - yy) * (7.f * zz - 1.f);
  B[23] = SH_C4[7] * xz * (xx - 3.f * yy);
  B[24] = SH_C4[8] * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
  return 25;
}

/* sh.cuh:188-205.  coeffs [n, K(degree), 3]; colors [n,3] */
GSR_API void gsr_oracle_sh_forward(int n, int degree, int degrees_to_use,
                                   const float *viewdirs, const float *coeffs,
                                   float *colors) {
  const int K = sh_bases(degree);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    float B[25];
    int nb = sh_basis_vec(degrees_to_use, viewdirs + 3 * i, B);
    const float *cf = coeffs + (size_t)i * K * 3;
    for (int c = 0; c < 3; ++c) {
      float acc = B[0] * cf[c];
      /* same band-by-band association as the kernel */
      if (nb >= 4) acc += (B[1] * cf[3 + c] + B[2] * cf[6 + c] + B[3] * cf[9 + c]);
      if (nb >= 9) {
        float t = 0.f;
        for (int k = 4; k < 9; ++k) t += B[k] * cf[3 * k + c];
        acc += t;
      }
      if (nb >= 16) {
        float t = 0.f;
        for (int k = 9; k < 16; ++k) t += B[k] * cf[3 * k + c];
        acc += t;
      }
      if (nb >= 25) {
        float t = 0.f;
        for (int k = 16; k < 25; ++k) t += B[k] * cf[3 * k + c];
        acc += t;
      }
      colors[3 * i + c] = acc;
    }
  }
}

/* sh.cuh:207-224.  v_coeffs [n,K,3] fully written (zeros above the bands used) */
GSR_API void gsr_oracle_sh_backward(int n, int degree, int degrees_to_use,
                                    const float *viewdirs,
                                    const float *v_colors, float *v_coeffs) {
  const int K = sh_bases(degree);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    float B[25];
    int nb = sh_basis_vec(degrees_to_use, viewdirs + 3 * i, B);
    float *vc = v_coeffs + (size_t)i * K * 3;
    for (int k = 0; k < K; ++k)
      for (int c = 0; c < 3; ++c)
        vc[3 * k + c] = (k < nb) ? B[k] * v_colors[3 * i + c] : 0.f;
  }
}

/* ------------------------------------------------------------- binning */

/* utils.py:106-125 : inclusive int32 scan; returns the total */
GSR_API int gsr_oracle_cumsum(int n, const int *num_tiles_hit, int *cum) {
  int acc = 0;
  for (int i = 0; i < n; ++i) {
    acc += num_tiles_hit[i];
    cum[i] = acc;
  }
  return acc;
}

/* forward.cu:94-127 */
GSR_API void gsr_oracle_map_intersects(int n, const float *xys,
                                       const float *depths, const int *radii,
                                       const int *cum_tiles_hit, int tiles_x,
                                       int tiles_y, int bw, int64_t *isect_ids,
                                       int *gaussian_ids) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int i = 0; i < n; ++i) {
    if (radii[i] <= 0) continue;
    int minx, miny, maxx, maxy;
    tile_bbox(xys[2 * i], xys[2 * i + 1], (float)radii[i], tiles_x, tiles_y,
              bw, &minx, &miny, &maxx, &maxy);
    int cur = (i == 0) ? 0 : cum_tiles_hit[i - 1];
    int32_t dbits;
    memcpy(&dbits, depths + i, 4);
    int64_t depth_id = (int64_t)dbits;
    for (int ty = miny; ty < maxy; ++ty)
      for (int tx = minx; tx < maxx; ++tx) {
        int64_t tile_id = (int64_t)ty * tiles_x + tx;
        isect_ids[cur] = (tile_id << 32) | depth_id;
        gaussian_ids[cur] = i;
        ++cur;
      }
  }
}

/* utils.py:179-180 : sort keys ascending (stable), permute values */
typedef struct { int64_t k; int v; } kv_t;
static void kv_merge_sort(kv_t *a, kv_t *tmp, long n) {
  for (long width = 1; width < n; width *= 2) {
#pragma omp parallel for schedule(static)
    for (long lo = 0; lo < n; lo += 2 * width) {
      long mid = lo + width < n ? lo + width : n;
      long hi = lo + 2 * width < n ? lo + 2 * width : n;
      long i = lo, j = mid, o = lo;
      while (i < mid && j < hi) tmp[o++] = (a[j].k < a[i].k) ? a[j++] : a[i++];
      while (i < mid) tmp[o++] = a[i++];
      while (j < hi) tmp[o++] = a[j++];
    }
    memcpy(a, tmp, (size_t)n * sizeof(kv_t));
  }
}
GSR_API void gsr_oracle_sort_intersects(int num_intersects,
                                        const int64_t *isect_ids,
                                        const int *gaussian_ids,
                                        int64_t *isect_ids_sorted,
                                        int *gaussian_ids_sorted) {
  long n = num_intersects;
  if (n <= 0) return;
  kv_t *a = (kv_t *)malloc((size_t)n * sizeof(kv_t));
  kv_t *t = (kv_t *)malloc((size_t)n * sizeof(kv_t));
  for (long i = 0; i < n; ++i) {
    a[i].k = isect_ids[i];
    a[i].v = gaussian_ids[i];
  }
  kv_merge_sort(a, t, n);
  for (long i = 0; i < n; ++i) {
    isect_ids_sorted[i] = a[i].k;
    gaussian_ids_sorted[i] = a[i].v;
  }
  free(a);
  free(t);
}

/* forward.cu:132-154 ; tile_bins [num_tiles,2] fully written (zero = empty) */
GSR_API void gsr_oracle_tile_bin_edges(int num_intersects,
                                       const int64_t *isect_ids_sorted,
                                       int num_tiles, int *tile_bins) {
  memset(tile_bins, 0, (size_t)num_tiles * 2 * sizeof(int));
  for (int i = 0; i < num_intersects; ++i) {
    int cur = (int)(isect_ids_sorted[i] >> 32);
    if (i == 0) tile_bins[2 * cur] = 0;
    if (i == num_intersects - 1) tile_bins[2 * cur + 1] = num_intersects;
    if (i == 0) continue;
    int prev = (int)(isect_ids_sorted[i - 1] >> 32);
    if (prev != cur) {
      tile_bins[2 * prev + 1] = i;
      tile_bins[2 * cur] = i;
    }
  }
}

/* --------------------------------------------------- rasterize forward */

/* forward.cu:278-395 (channels==3) and 159-276 (channels!=3, here fp32
 * accumulators).  `ambig` (optional, may be NULL) is set to 1 for pixels where
 * a discrete decision (sigma<0, alpha<1/255, T(1-alpha)<=1e-4) is within
 * `ambig_eps` (relative) of flipping -- such pixels are legitimately unstable
 * under 1-ulp differences in exp() and are compared at a looser tolerance. */
GSR_API void gsr_oracle_rasterize_forward(
    int tiles_x, int tiles_y, int bw, int img_w, int img_h, int channels,
    const int *gaussian_ids_sorted, const int *tile_bins, const float *xys,
    const float *conics, const float *colors, const float *opacities,
    const float *background, float *out_img, float *final_Ts, int *final_idx,
    unsigned char *ambig, float ambig_eps) {
  (void)tiles_y;
#pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < img_h; ++i) {
    float *acc = (float *)malloc(sizeof(float) * (size_t)channels);
    for (int j = 0; j < img_w; ++j) {
      int tile = (i / bw) * tiles_x + (j / bw);
      int lo = tile_bins[2 * tile], hi = tile_bins[2 * tile + 1];
      float px = (float)j, py = (float)i;
      float T = 1.f;
      int cur_idx = 0;
      unsigned char amb = 0;
      for (int c = 0; c < channels; ++c) acc[c] = 0.f;
      for (int idx = lo; idx < hi; ++idx) {
        int g = gaussian_ids_sorted[idx];
        float dx = xys[2 * g] - px, dy = xys[2 * g + 1] - py;
        float a = conics[3 * g], b = conics[3 * g + 1], cc = conics[3 * g + 2];
        float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
        float alpha = fminf_(0.999f, opacities[g] * expf(-sigma));
        float stol = 0.f; /* how far sigma may move under a different fp32 evaluation order */
        if (ambig) {
          stol = ambig_eps + 1e-6f * (0.5f * (fabsf(a) * dx * dx + fabsf(cc) * dy * dy) + fabsf(b * dx * dy));
          if (fabsf(sigma) <= stol) amb = 1;
          if (fabsf(alpha - 1.f / 255.f) <= stol * (1.f / 255.f)) amb = 1;
        }
        if (sigma < 0.f || alpha < 1.f / 255.f) continue;
        float next_T = T * (1.f - alpha);
        if (ambig && fabsf(next_T - 1e-4f) <= (20.f * ambig_eps + stol) * 1e-4f) amb = 1;
        if (next_T <= 1e-4f) break;
        float vis = alpha * T;
        for (int c = 0; c < channels; ++c)
          acc[c] += colors[(size_t)channels * g + c] * vis;
        T = next_T;
        cur_idx = idx;
      }
      size_t pid = (size_t)i * img_w + j;
      final_Ts[pid] = T;
      final_idx[pid] = cur_idx;
      for (int c = 0; c < channels; ++c)
        out_img[pid * channels + c] = acc[c] + T * background[c];
      if (ambig) ambig[pid] = amb;
    }
    free(acc);
  }
}

/* -------------------------------------------------- rasterize backward */

/* backward.cu:133-303 (and 23-131 for channels!=3, fp32 running sum here).
 * Per-Gaussian gradients are accumulated in double per thread and reduced in
 * a fixed order; outputs are fully written. */
GSR_API void gsr_oracle_rasterize_backward(
    int img_h, int img_w, int bw, int channels, int num_points,
    const int *gaussian_ids_sorted, const int *tile_bins, const float *xys,
    const float *conics, const float *colors, const float *opacities,
    const float *background, const float *final_Ts, const int *final_idx,
    const float *v_output, const float *v_output_alpha, float *v_xy,
    float *v_conic, float *v_colors, float *v_opacity, float *abs_sums,
    unsigned char *ambig, float ambig_eps) {
  /* ambig (optional, [num_points]): 1 for Gaussians with a pixel whose skip
   * decision (sigma<0, alpha<1/255) is within ambig_eps (relative) of flipping */
  /* abs_sums (optional, [num_points, 6+channels], order xy(2) conic(3)
   * opacity(1) colors(C)): sum of |per-pixel term| of every gradient component
   * -- the scale fp32 accumulation error is relative to (tests use it to bound
   * the error of sums with heavy cancellation). */
  const int tiles_x = (img_w + bw - 1) / bw;
  const int stride = 6 + channels; /* xy(2) conic(3) opac(1) colors(C) */
  const int want_abs = abs_sums != NULL;
  if (ambig) memset(ambig, 0, (size_t)num_points);
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  /* bound the scratch: fall back to fewer accumulators for huge N */
  while (nthreads > 1 && (double)nthreads * num_points * stride * 8.0 > 6e9) nthreads /= 2;
  double *accs = (double *)calloc((size_t)nthreads * num_points * stride, sizeof(double));
  double *absacc = want_abs ? (double *)calloc((size_t)nthreads * num_points * stride, sizeof(double)) : NULL;

#pragma omp parallel num_threads(nthreads)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double *A = accs + (size_t)tid * num_points * stride;
    double *AA = want_abs ? absacc + (size_t)tid * num_points * stride : NULL;
    float *buf = (float *)malloc(sizeof(float) * (size_t)channels);
#pragma omp for schedule(dynamic, 4)
    for (int i = 0; i < img_h; ++i) {
      for (int j = 0; j < img_w; ++j) {
        int tile = (i / bw) * tiles_x + (j / bw);
        int lo = tile_bins[2 * tile];
        size_t pid = (size_t)i * img_w + j;
        float px = (float)j, py = (float)i;
        float T_final = final_Ts[pid];
        float T = T_final;
        int bin_final = final_idx[pid];
        const float *vout = v_output + pid * channels;
        float vout_alpha = v_output_alpha[pid];
        for (int c = 0; c < channels; ++c) buf[c] = 0.f;
        int hi = tile_bins[2 * tile + 1];
        int start = bin_final < hi - 1 ? bin_final : hi - 1;
        for (int idx = start; idx >= lo; --idx) {
          int g = gaussian_ids_sorted[idx];
          float dx = xys[2 * g] - px, dy = xys[2 * g + 1] - py;
          float a = conics[3 * g], b = conics[3 * g + 1], cc = conics[3 * g + 2];
          float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
          float opac = opacities[g];
          float vis = expf(-sigma);
          float alpha = fminf_(0.99f, opac * vis);
          if (ambig) {
            float stol = ambig_eps + 1e-6f * (0.5f * (fabsf(a) * dx * dx + fabsf(cc) * dy * dy) + fabsf(b * dx * dy));
            if (fabsf(sigma) <= stol || fabsf(alpha - 1.f / 255.f) <= stol * (1.f / 255.f)) ambig[g] = 1;
          }
          if (sigma < 0.f || alpha < 1.f / 255.f) continue;
          float ra = 1.f / (1.f - alpha);
          T *= ra;
          float fac = alpha * T;
          float v_alpha = 0.f;
          double *Ag = A + (size_t)g * stride;
          const float *rgb = colors + (size_t)channels * g;
          for (int c = 0; c < channels; ++c) {
            Ag[6 + c] += (double)(fac * vout[c]);
            v_alpha += (rgb[c] * T - buf[c] * ra) * vout[c];
          }
          v_alpha += T_final * ra * vout_alpha;
          for (int c = 0; c < channels; ++c)
            v_alpha += -T_final * ra * background[c] * vout[c];
          for (int c = 0; c < channels; ++c) buf[c] += rgb[c] * fac;
          float v_sigma = -opac * vis * v_alpha;
          Ag[2] += (double)(0.5f * v_sigma * dx * dx);
          Ag[3] += (double)(v_sigma * dx * dy);
          Ag[4] += (double)(0.5f * v_sigma * dy * dy);
          Ag[0] += (double)(v_sigma * (a * dx + b * dy));
          Ag[1] += (double)(v_sigma * (b * dx + cc * dy));
          Ag[5] += (double)(vis * v_alpha);
          if (want_abs) {
            double *Bg = AA + (size_t)g * stride;
            Bg[0] += fabs((double)(v_sigma * (a * dx + b * dy)));
            Bg[1] += fabs((double)(v_sigma * (b * dx + cc * dy)));
            Bg[2] += fabs((double)(0.5f * v_sigma * dx * dx));
            Bg[3] += fabs((double)(v_sigma * dx * dy));
            Bg[4] += fabs((double)(0.5f * v_sigma * dy * dy));
            Bg[5] += fabs((double)(vis * v_alpha));
            for (int c = 0; c < channels; ++c) Bg[6 + c] += fabs((double)(fac * vout[c]));
          }
        }
      }
    }
    free(buf);
  }
#pragma omp parallel for schedule(static)
  for (int g = 0; g < num_points; ++g) {
    const int ns = stride;
    /* component k of Gaussian g: the threads' partial sums added in thread order */
    for (int k = 0; k < ns; ++k) {
      double sk = 0.0;
      for (int t = 0; t < nthreads; ++t) sk += accs[((size_t)t * num_points + g) * stride + k];
      const float f = (float)sk;
      if (k < 2) v_xy[2 * g + k] = f;
      else if (k < 5) v_conic[3 * g + (k - 2)] = f;
      else if (k == 5) v_opacity[g] = f;
      else v_colors[(size_t)channels * g + (k - 6)] = f;
    }
    if (want_abs) {
      for (int k = 0; k < ns; ++k) {
        double t = 0.0;
        for (int th = 0; th < nthreads; ++th) t += absacc[((size_t)th * num_points + g) * stride + k];
        abs_sums[(size_t)g * stride + k] = (float)t;
      }
    }
  }
  free(accs);
  free(absacc);
}

/* bindings.cu:19-56 : standalone conic + radius from cov2d */
GSR_API void gsr_oracle_cov2d_bounds(int n, const float *cov2d, float *conics,
                                     float *radii) {
  for (int i = 0; i < n; ++i) {
    float conic[3] = {0.f, 0.f, 0.f}, radius = 0.f;
    /* the kernel ignores the ok flag and stores whatever was computed */
    cov2d_bounds(cov2d + 3 * i, conic, &radius);
    conics[3 * i] = conic[0];
    conics[3 * i + 1] = conic[1];
    conics[3 * i + 2] = conic[2];
    radii[i] = radius;
  }
}

/* ------------------------------------------------- loss head (row f2) */

/* (1-lambda)*mean|x-y| + lambda*(1 - SSIM(x,y)) and d/dx, for [H,W,3] images.
 * Restates the call site gs_toolkit/models/vanilla_gs.py:926-944 and the
 * published algorithm of pytorch_msssim 1.0.0 SSIM(data_range=1, channel=3,
 * size_average=True): 11-tap Gaussian (sigma 1.5), valid padding, C1=0.01^2,
 * C2=0.03^2.  Double precision throughout.  v_pred may be NULL. */
GSR_API double gsr_oracle_l1_ssim(int H, int W, const float *pred, const float *gt,
                                  float ssim_lambda, double *out_l1, double *out_ssim,
                                  float *v_pred) {
  double w[11], wsum = 0.0;
  for (int k = 0; k < 11; ++k) { w[k] = exp(-((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); wsum += w[k]; }
  for (int k = 0; k < 11; ++k) w[k] /= wsum;
  const int Hv = H - 10, Wv = W - 10;
  const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
  double l1 = 0.0, ss = 0.0;
  const size_t np = (size_t)H * W * 3;
  for (size_t i = 0; i < np; ++i) l1 += fabs((double)pred[i] - (double)gt[i]);
  l1 /= (double)np;
  double *Dm = NULL, *D11 = NULL, *D12 = NULL;
  if (v_pred) {
    Dm = (double *)calloc((size_t)Hv * Wv * 3, sizeof(double));
    D11 = (double *)calloc((size_t)Hv * Wv * 3, sizeof(double));
    D12 = (double *)calloc((size_t)Hv * Wv * 3, sizeof(double));
  }
#pragma omp parallel for reduction(+ : ss) schedule(static)
  for (int i = 0; i < Hv; ++i)
    for (int j = 0; j < Wv; ++j)
      for (int c = 0; c < 3; ++c) {
        double mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
        for (int k = 0; k < 11; ++k)
          for (int l = 0; l < 11; ++l) {
            const size_t o = ((size_t)(i + k) * W + (j + l)) * 3 + c;
            const double x = pred[o], y = gt[o], ww = w[k] * w[l];
            mu1 += ww * x; mu2 += ww * y; e11 += ww * x * x; e22 += ww * y * y; e12 += ww * x * y;
          }
        const double s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const double A1 = 2 * mu1 * mu2 + C1, A2 = 2 * s12 + C2;
        const double B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
        const double S = A1 * A2 / (B1 * B2);
        ss += S;
        if (v_pred) {
          const size_t m = ((size_t)i * Wv + j) * 3 + c;
          const double dA1 = A2 / (B1 * B2), dA2 = A1 / (B1 * B2), dB1 = -S / B1, dB2 = -S / B2;
          Dm[m] = dA1 * 2 * mu2 + dB1 * 2 * mu1 - dA2 * 2 * mu2 - dB2 * 2 * mu1;
          D11[m] = dB2;
          D12[m] = 2 * dA2;
        }
      }
  ss /= (double)Hv * Wv * 3;
  if (v_pred) {
    const double kl1 = (1.0 - ssim_lambda) / (double)np;
    const double kss = -(double)ssim_lambda / ((double)Hv * Wv * 3);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < W; ++j)
        for (int c = 0; c < 3; ++c) {
          const size_t o = ((size_t)i * W + j) * 3 + c;
          const double x = pred[o], y = gt[o];
          double g = 0.0;
          for (int k = 0; k < 11; ++k) {
            const int mi = i - k;
            if (mi < 0 || mi >= Hv) continue;
            for (int l = 0; l < 11; ++l) {
              const int mj = j - l;
              if (mj < 0 || mj >= Wv) continue;
              const size_t m = ((size_t)mi * Wv + mj) * 3 + c;
              g += w[k] * w[l] * (Dm[m] + 2 * x * D11[m] + y * D12[m]);
            }
          }
          const double d = x - y;
          v_pred[o] = (float)(kl1 * (d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0)) + kss * g);
        }
    free(Dm); free(D11); free(D12);
  }
  if (out_l1) *out_l1 = l1;
  if (out_ssim) *out_ssim = ss;
  return (1.0 - ssim_lambda) * l1 + ssim_lambda * (1.0 - ss);
}
