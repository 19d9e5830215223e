// project.hip -- EWA projection of 3-D Gaussians and its VJP (gfx950).
//
// One lane per Gaussian; purely streaming (100 B in+out per Gaussian forward,
// ~190 B backward), HBM-bound.  The camera matrices are wave-uniform kernel
// arguments read through the scalar cache.  All outputs are written by every
// lane (zeros for culled splats) so callers can hand in uninitialised buffers.
//
// Behaviour restated from the reference kernels (gs_toolkit/gs_components/
// rasterizer/cuda/csrc): forward.cu:13-90,398-464, backward.cu:305-453,
// helpers.cuh:7-219.  3x3 algebra is written out on row-major scalars.
//
// Parity: this file is compiled with -ffp-contract=off (Makefile), like the CPU oracle
// (oracle/Makefile), and evaluates every expression in the oracle's operation order with
// correctly rounded division and square root (hipcc's default): the forward outputs --
// including the integer radii / num_tiles_hit, which sit behind a ceil() -- are then
// bit-identical to the oracle's (tests/test_gpu_kernels.py::test_project_forward).  The
// kernels are HBM-bound, so the few un-fused multiply-adds cost nothing measurable.
#include "gsr_common.h"

namespace {

struct M3 {
  float a00, a01, a02, a10, a11, a12, a20, a21, a22;
};

__device__ __forceinline__ M3 mul(const M3 &A, const M3 &B) {
  M3 C;
  C.a00 = A.a00 * B.a00 + A.a01 * B.a10 + A.a02 * B.a20;
  C.a01 = A.a00 * B.a01 + A.a01 * B.a11 + A.a02 * B.a21;
  C.a02 = A.a00 * B.a02 + A.a01 * B.a12 + A.a02 * B.a22;
  C.a10 = A.a10 * B.a00 + A.a11 * B.a10 + A.a12 * B.a20;
  C.a11 = A.a10 * B.a01 + A.a11 * B.a11 + A.a12 * B.a21;
  C.a12 = A.a10 * B.a02 + A.a11 * B.a12 + A.a12 * B.a22;
  C.a20 = A.a20 * B.a00 + A.a21 * B.a10 + A.a22 * B.a20;
  C.a21 = A.a20 * B.a01 + A.a21 * B.a11 + A.a22 * B.a21;
  C.a22 = A.a20 * B.a02 + A.a21 * B.a12 + A.a22 * B.a22;
  return C;
}
__device__ __forceinline__ M3 transpose(const M3 &A) {
  return M3{A.a00, A.a10, A.a20, A.a01, A.a11, A.a21, A.a02, A.a12, A.a22};
}

// (w,x,y,z) quaternion -> rotation; renormalises (helpers.cuh:144-159)
__device__ __forceinline__ M3 quat_to_rot(float qw, float qx, float qy, float qz,
                                          float &w, float &x, float &y, float &z) {
  const float s = 1.f / sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);  // (not rsqrtf: see the parity note)
  w = qw * s;
  x = qx * s;
  y = qy * s;
  z = qz * s;
  M3 R;
  R.a00 = 1.f - 2.f * (y * y + z * z);
  R.a01 = 2.f * (x * y - w * z);
  R.a02 = 2.f * (x * z + w * y);
  R.a10 = 2.f * (x * y + w * z);
  R.a11 = 1.f - 2.f * (x * x + z * z);
  R.a12 = 2.f * (y * z - w * x);
  R.a20 = 2.f * (x * z - w * y);
  R.a21 = 2.f * (y * z + w * x);
  R.a22 = 1.f - 2.f * (x * x + y * y);
  return R;
}

struct Cam {
  float v[12];  // view matrix, top 3x4, row-major
  float p[16];  // full projection (P*V), row-major
};

__global__ __launch_bounds__(256) void project_fwd_kernel(
    const int n, const float *__restrict__ means3d,
    const float *__restrict__ scales, const float glob_scale,
    const float *__restrict__ quats, const float *__restrict__ viewmat,
    const float *__restrict__ projmat, const float fx, const float fy,
    const float cx, const float cy, const int img_w, const int img_h,
    const int tiles_x, const int tiles_y, const int bw, const float clip,
    float *__restrict__ cov3d, float *__restrict__ xys,
    float *__restrict__ depths, int *__restrict__ radii,
    float *__restrict__ conics, float *__restrict__ compensation,
    int *__restrict__ num_tiles_hit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;

  float o_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float o_x = 0.f, o_y = 0.f, o_depth = 0.f, o_comp = 0.f;
  float o_k0 = 0.f, o_k1 = 0.f, o_k2 = 0.f;
  int o_radius = 0, o_tiles = 0;

  const float px = means3d[3 * i], py = means3d[3 * i + 1], pz = means3d[3 * i + 2];
  const float *V = viewmat;
  const float tx = V[0] * px + V[1] * py + V[2] * pz + V[3];
  const float ty = V[4] * px + V[5] * py + V[6] * pz + V[7];
  const float tz = V[8] * px + V[9] * py + V[10] * pz + V[11];

  if (tz > clip) {  // near-plane cull is `z <= clip` (helpers.cuh:212-219)
    M3 S;
    if (scales) {
      float w, x, y, z;
      const M3 R = quat_to_rot(quats[4 * i], quats[4 * i + 1], quats[4 * i + 2],
                               quats[4 * i + 3], w, x, y, z);
      const float s0 = glob_scale * scales[3 * i], s1 = glob_scale * scales[3 * i + 1],
                  s2 = glob_scale * scales[3 * i + 2];
      const M3 M{R.a00 * s0, R.a01 * s1, R.a02 * s2, R.a10 * s0, R.a11 * s1,
                 R.a12 * s2, R.a20 * s0, R.a21 * s1, R.a22 * s2};
      S = mul(M, transpose(M));
    } else {  // precomputed covariances: cov3d is an INPUT (upper triangle xx xy xz yy yz zz)
      const float *c = cov3d + 6 * i;
      S = M3{c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
    }
    o_cov[0] = S.a00;
    o_cov[1] = S.a01;
    o_cov[2] = S.a02;
    o_cov[3] = S.a11;
    o_cov[4] = S.a12;
    o_cov[5] = S.a22;

    // EWA: clamp the centre to 1.3x the frustum, J = d(pix)/d(t)
    const float limx = 1.3f * (0.5f * (float)img_w / fx);
    const float limy = 1.3f * (0.5f * (float)img_h / fy);
    const float ex = tz * fminf(limx, fmaxf(-limx, tx / tz));
    const float ey = tz * fminf(limy, fmaxf(-limy, ty / tz));
    const float rz = 1.f / tz, rz2 = rz * rz;
    const M3 J{fx * rz, 0.f, -fx * ex * rz2, 0.f, fy * rz, -fy * ey * rz2, 0.f, 0.f, 0.f};
    const M3 W{V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
    const M3 T = mul(J, W);
    const M3 Vs{S.a00, S.a01, S.a02, S.a01, S.a11, S.a12, S.a02, S.a12, S.a22};
    const M3 C = mul(mul(T, Vs), transpose(T));
    const float det_orig = C.a00 * C.a11 - C.a01 * C.a01;
    const float c0 = C.a00 + 0.3f, c1 = C.a01, c2 = C.a11 + 0.3f;
    const float det_blur = c0 * c2 - c1 * c1;
    const float comp = sqrtf(fmaxf(0.f, det_orig / det_blur));

    float k0, k1, k2, radius;
    if (gsr_cov2d_bounds(c0, c1, c2, k0, k1, k2, radius)) {
      o_k0 = k0;
      o_k1 = k1;
      o_k2 = k2;
      const float *P = projmat;
      const float hx = P[0] * px + P[1] * py + P[2] * pz + P[3];
      const float hy = P[4] * px + P[5] * py + P[6] * pz + P[7];
      const float hw = P[12] * px + P[13] * py + P[14] * pz + P[15];
      const float rw = 1.f / (hw + 1e-6f);
      const float u = 0.5f * (float)img_w * (hx * rw) + cx - 0.5f;
      const float v = 0.5f * (float)img_h * (hy * rw) + cy - 0.5f;
      int minx, miny, maxx, maxy;
      gsr_tile_bbox(u, v, radius, tiles_x, tiles_y, 0.f, bw, minx, miny, maxx, maxy);
      const int area = (maxx - minx) * (maxy - miny);
      if (area > 0) {
        o_tiles = area;
        o_depth = tz;
        o_radius = (int)radius;
        o_x = u;
        o_y = v;
        o_comp = comp;
      }
    }
  }

  if (scales) {
#pragma unroll
    for (int k = 0; k < 6; ++k) gsr_store_stream(cov3d + 6 * i + k, o_cov[k]);  // (read again by the backward only)
  }
  xys[2 * i] = o_x;
  xys[2 * i + 1] = o_y;
  depths[i] = o_depth;
  radii[i] = o_radius;
  conics[3 * i] = o_k0;
  conics[3 * i + 1] = o_k1;
  conics[3 * i + 2] = o_k2;
  gsr_store_stream(compensation + i, o_comp);
  num_tiles_hit[i] = o_tiles;
}

__global__ __launch_bounds__(256) void project_bwd_kernel(
    const int n, const float *__restrict__ means3d,
    const float *__restrict__ scales, const float glob_scale,
    const float *__restrict__ quats, const float *__restrict__ viewmat,
    const float *__restrict__ projmat, const float fx, const float fy,
    const int img_w, const int img_h, const float *__restrict__ cov3d,
    const int *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ compensation, const float *__restrict__ v_xy,
    const float *__restrict__ v_depth, const float *__restrict__ v_conic,
    const float *__restrict__ v_compensation, float *__restrict__ v_cov2d,
    float *__restrict__ v_cov3d, float *__restrict__ v_mean3d,
    float *__restrict__ v_scale, float *__restrict__ v_quat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;

  float g2[3] = {0.f, 0.f, 0.f};
  float g3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float gm[3] = {0.f, 0.f, 0.f};
  float gs[3] = {0.f, 0.f, 0.f};
  float gq[4] = {0.f, 0.f, 0.f, 0.f};

  if (radii[i] > 0) {
    const float px = means3d[3 * i], py = means3d[3 * i + 1], pz = means3d[3 * i + 2];
    const float *P = projmat;
    const float *V = viewmat;

    // pixel = ndc2pix(P p / (w + eps))  ->  d/dp  (helpers.cuh:125-142)
    const float hx = P[0] * px + P[1] * py + P[2] * pz + P[3];
    const float hy = P[4] * px + P[5] * py + P[6] * pz + P[7];
    const float hw = P[12] * px + P[13] * py + P[14] * pz + P[15];
    const float rw = 1.f / (hw + 1e-6f);
    const float vnx = v_xy ? 0.5f * (float)img_w * v_xy[2 * i] : 0.f;
    const float vny = v_xy ? 0.5f * (float)img_h * v_xy[2 * i + 1] : 0.f;
    const float vt0 = vnx * rw, vt1 = vny * rw;
    const float vt3 = -(vnx * hx + vny * hy) * rw * rw;
    gm[0] = P[0] * vt0 + P[4] * vt1 + P[12] * vt3;
    gm[1] = P[1] * vt0 + P[5] * vt1 + P[13] * vt3;
    gm[2] = P[2] * vt0 + P[6] * vt1 + P[14] * vt3;

    // depth = V[2,:] . p
    const float vz = v_depth ? v_depth[i] : 0.f;
    gm[0] += V[8] * vz;
    gm[1] += V[9] * vz;
    gm[2] += V[10] * vz;

    // conic = inv(cov2d)  ->  v_cov2d = -X G X  (helpers.cuh:62-74)
    const float X00 = conics[3 * i], X01 = conics[3 * i + 1], X11 = conics[3 * i + 2];
    const float G00 = v_conic ? v_conic[3 * i] : 0.f, G01 = v_conic ? 0.5f * v_conic[3 * i + 1] : 0.f,
                G11 = v_conic ? v_conic[3 * i + 2] : 0.f;
    const float A00 = X00 * G00 + X01 * G01, A01 = X00 * G01 + X01 * G11;
    const float A10 = X01 * G00 + X11 * G01, A11 = X01 * G01 + X11 * G11;
    g2[0] = -(A00 * X00 + A01 * X01);
    g2[1] = -(A00 * X01 + A01 * X11) - (A10 * X00 + A11 * X01);
    g2[2] = -(A10 * X01 + A11 * X11);

    // compensation = sqrt(det(cov2d - 0.3 I) / det(cov2d))  (helpers.cuh:76-90)
    {
      const float comp = compensation[i];
      const float inv_det = X00 * X11 - X01 * X01;
      const float om2 = 1.f - comp * comp;
      const float vsq = (v_compensation ? v_compensation[i] : 0.f) * 0.5f / (comp + 1e-6f);
      g2[0] += vsq * (om2 * X00 - 0.3f * inv_det);
      g2[1] += 2.f * vsq * (om2 * X01);
      g2[2] += vsq * (om2 * X11 - 0.3f * inv_det);
    }

    // cov2d = T V T^T, T = J W  (backward.cu:350-423; no fov clamp here)
    const M3 W{V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
    const float tx = V[0] * px + V[1] * py + V[2] * pz + V[3];
    const float ty = V[4] * px + V[5] * py + V[6] * pz + V[7];
    const float tz = V[8] * px + V[9] * py + V[10] * pz + V[11];
    const float rz = 1.f / tz, rz2 = rz * rz, rz3 = rz2 * rz;
    const M3 J{fx * rz, 0.f, -fx * tx * rz2, 0.f, fy * rz, -fy * ty * rz2, 0.f, 0.f, 0.f};
    const float *c3 = cov3d + 6 * i;
    const M3 Vs{c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
    const M3 Gc{g2[0], 0.5f * g2[1], 0.f, 0.5f * g2[1], g2[2], 0.f, 0.f, 0.f, 0.f};
    const M3 T = mul(J, W);
    const M3 vV = mul(mul(transpose(T), Gc), T);
    g3[0] = vV.a00;
    g3[1] = vV.a01 + vV.a10;
    g3[2] = vV.a02 + vV.a20;
    g3[3] = vV.a11;
    g3[4] = vV.a12 + vV.a21;
    g3[5] = vV.a22;
    // v_T = G T V^T + G^T T V
    const M3 P1 = mul(mul(Gc, T), transpose(Vs));
    const M3 P2 = mul(mul(transpose(Gc), T), Vs);
    const M3 vT{P1.a00 + P2.a00, P1.a01 + P2.a01, P1.a02 + P2.a02,
                P1.a10 + P2.a10, P1.a11 + P2.a11, P1.a12 + P2.a12,
                P1.a20 + P2.a20, P1.a21 + P2.a21, P1.a22 + P2.a22};
    const M3 vJ = mul(vT, transpose(W));
    const float vtx = -fx * rz2 * vJ.a02;
    const float vty = -fy * rz2 * vJ.a12;
    const float vtz = -fx * rz2 * vJ.a00 + 2.f * fx * tx * rz3 * vJ.a02 -
                      fy * rz2 * vJ.a11 + 2.f * fy * ty * rz3 * vJ.a12;
    gm[0] += vtx * W.a00 + vty * W.a10 + vtz * W.a20;
    gm[1] += vtx * W.a01 + vty * W.a11 + vtz * W.a21;
    gm[2] += vtx * W.a02 + vty * W.a12 + vtz * W.a22;

    // cov3d = M M^T, M = R S  (backward.cu:427-453)
    const M3 vS{g3[0], 0.5f * g3[1], 0.5f * g3[2], 0.5f * g3[1], g3[3],
                0.5f * g3[4], 0.5f * g3[2], 0.5f * g3[4], g3[5]};
    if (scales) {  // (precomputed covariances: the chain ends at v_cov3d)
    float w, x, y, z;
    const M3 R = quat_to_rot(quats[4 * i], quats[4 * i + 1], quats[4 * i + 2],
                             quats[4 * i + 3], w, x, y, z);
    const float s0 = glob_scale * scales[3 * i], s1 = glob_scale * scales[3 * i + 1],
                s2 = glob_scale * scales[3 * i + 2];
    const M3 M{R.a00 * s0, R.a01 * s1, R.a02 * s2, R.a10 * s0, R.a11 * s1,
               R.a12 * s2, R.a20 * s0, R.a21 * s1, R.a22 * s2};
    M3 vM = mul(vS, M);
    vM.a00 *= 2.f; vM.a01 *= 2.f; vM.a02 *= 2.f;
    vM.a10 *= 2.f; vM.a11 *= 2.f; vM.a12 *= 2.f;
    vM.a20 *= 2.f; vM.a21 *= 2.f; vM.a22 *= 2.f;
    gs[0] = (R.a00 * vM.a00 + R.a10 * vM.a10 + R.a20 * vM.a20) * glob_scale;
    gs[1] = (R.a01 * vM.a01 + R.a11 * vM.a11 + R.a21 * vM.a21) * glob_scale;
    gs[2] = (R.a02 * vM.a02 + R.a12 * vM.a12 + R.a22 * vM.a22) * glob_scale;
    const M3 vR{vM.a00 * s0, vM.a01 * s1, vM.a02 * s2, vM.a10 * s0, vM.a11 * s1,
                vM.a12 * s2, vM.a20 * s0, vM.a21 * s1, vM.a22 * s2};
    // d(R)/d(q) with q treated as unit (helpers.cuh:161-200)
    gq[0] = 2.f * (x * (vR.a21 - vR.a12) + y * (vR.a02 - vR.a20) + z * (vR.a10 - vR.a01));
    gq[1] = 2.f * (-2.f * x * (vR.a11 + vR.a22) + y * (vR.a10 + vR.a01) +
                   z * (vR.a20 + vR.a02) + w * (vR.a21 - vR.a12));
    gq[2] = 2.f * (x * (vR.a10 + vR.a01) - 2.f * y * (vR.a00 + vR.a22) +
                   z * (vR.a21 + vR.a12) + w * (vR.a02 - vR.a20));
    gq[3] = 2.f * (x * (vR.a20 + vR.a02) + y * (vR.a21 + vR.a12) -
                   2.f * z * (vR.a00 + vR.a11) + w * (vR.a10 - vR.a01));
    }
  }

#pragma unroll
  for (int k = 0; k < 3; ++k) gsr_store_stream(v_cov2d + 3 * i + k, g2[k]);
#pragma unroll
  for (int k = 0; k < 6; ++k) gsr_store_stream(v_cov3d + 6 * i + k, g3[k]);  // (scratch of the chain: nobody reads it)
#pragma unroll
  for (int k = 0; k < 3; ++k) gsr_store_stream(v_mean3d + 3 * i + k, gm[k]);
  if (scales) {
#pragma unroll
    for (int k = 0; k < 3; ++k) v_scale[3 * i + k] = gs[k];  // (read next by the activation backward)
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quat[4 * i + k] = gq[k];
  }
}

__global__ __launch_bounds__(256) void cov2d_bounds_kernel(
    const int n, const float *__restrict__ cov2d, float *__restrict__ conics,
    float *__restrict__ radii) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float k0 = 0.f, k1 = 0.f, k2 = 0.f, r = 0.f;
  gsr_cov2d_bounds(cov2d[3 * i], cov2d[3 * i + 1], cov2d[3 * i + 2], k0, k1, k2, r);
  conics[3 * i] = k0;
  conics[3 * i + 1] = k1;
  conics[3 * i + 2] = k2;
  radii[i] = r;
}

}  // namespace

GSR_EXPORT int gsr_project_forward(
    int num_points, const float *means3d, const float *scales, float glob_scale,
    const float *quats, const float *viewmat, const float *projmat, float fx,
    float fy, float cx, float cy, unsigned img_height, unsigned img_width,
    unsigned block_width, float clip_thresh, float *cov3d, float *xys,
    float *depths, int32_t *radii, float *conics, float *compensation,
    int32_t *num_tiles_hit, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "project_forward: num_points < 0");
  GSR_REQUIRE(block_width >= 2 && block_width <= 16, "project_forward: block_width must be in [2,16]");
  GSR_REQUIRE(img_height > 0 && img_width > 0, "project_forward: empty image");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(means3d && viewmat && projmat && cov3d && xys && depths &&
                  radii && conics && compensation && num_tiles_hit,
              "project_forward: null pointer");
  GSR_REQUIRE((scales == nullptr) == (quats == nullptr),
              "project_forward: pass both scales and quats, or neither (cov3d is then an input)");
  const int tiles_x = (int)gsr_cdiv(img_width, block_width);
  const int tiles_y = (int)gsr_cdiv(img_height, block_width);
  hipLaunchKernelGGL(project_fwd_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0,
                     (hipStream_t)stream, num_points, means3d, scales, glob_scale, quats,
                     viewmat, projmat, fx, fy, cx, cy, (int)img_width, (int)img_height,
                     tiles_x, tiles_y, (int)block_width, clip_thresh, cov3d, xys, depths,
                     radii, conics, compensation, num_tiles_hit);
  GSR_CHECK_LAUNCH("project_forward");
  return GSR_OK;
}

GSR_EXPORT int gsr_project_backward(
    int num_points, const float *means3d, const float *scales, float glob_scale,
    const float *quats, const float *viewmat, const float *projmat, float fx,
    float fy, float cx, float cy, unsigned img_height, unsigned img_width,
    const float *cov3d, const int32_t *radii, const float *conics,
    const float *compensation, const float *v_xy, const float *v_depth,
    const float *v_conic, const float *v_compensation, float *v_cov2d,
    float *v_cov3d, float *v_mean3d, float *v_scale, float *v_quat,
    gsr_stream_t stream) {
  (void)cx;
  (void)cy;
  GSR_REQUIRE(num_points >= 0, "project_backward: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(means3d && viewmat && projmat && cov3d && radii && conics && compensation && v_cov2d &&
                  v_cov3d && v_mean3d,
              "project_backward: null pointer");
  GSR_REQUIRE((scales == nullptr) == (quats == nullptr), "project_backward: pass both scales and quats, or neither");
  GSR_REQUIRE(scales == nullptr || (v_scale && v_quat), "project_backward: null pointer");
  hipLaunchKernelGGL(project_bwd_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0,
                     (hipStream_t)stream, num_points, means3d, scales, glob_scale, quats,
                     viewmat, projmat, fx, fy, (int)img_width, (int)img_height, cov3d, radii,
                     conics, compensation, v_xy, v_depth, v_conic, v_compensation, v_cov2d,
                     v_cov3d, v_mean3d, v_scale, v_quat);
  GSR_CHECK_LAUNCH("project_backward");
  return GSR_OK;
}

GSR_EXPORT int gsr_cov2d_bounds(int num_pts, const float *cov2d, float *conics,
                                float *radii, gsr_stream_t stream) {
  GSR_REQUIRE(num_pts >= 0, "cov2d_bounds: num_pts < 0");
  if (num_pts == 0) return GSR_OK;
  GSR_REQUIRE(cov2d && conics && radii, "cov2d_bounds: null pointer");
  hipLaunchKernelGGL(cov2d_bounds_kernel, dim3(gsr_cdiv(num_pts, 256)), dim3(256), 0,
                     (hipStream_t)stream, num_pts, cov2d, conics, radii);
  GSR_CHECK_LAUNCH("cov2d_bounds");
  return GSR_OK;
}
