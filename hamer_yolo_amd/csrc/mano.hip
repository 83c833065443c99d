// rot6d_to_rotmat (geometry.py:47-70) + MANO forward (mano_wrapper.py:32-44 -> smplx.lbs.lbs,
// same arithmetic as the in-tree manopth layer, manolayer.py:172-262) + camera translation and
// 2-D projection (hamer.py:131-154), fused.  L2-bound fp32 work, ~1 MFLOP per hand, dominated by the
// 1.26 MB of pose blend shapes: NCHUNK workgroups per hand, each repeats the cheap shape / joint /
// kinematic-chain part (everything per-hand lives in LDS, 22 KB) and then poses and skins only its
// quarter of the vertices, so a workgroup streams 0.3 MB of posedirs instead of all of it and a
// batch of 64 hands fills the 256 CUs (one workgroup per hand: 150 us; this form: see DESIGN.md).
#include "common.h"
#include "hamer_hip_internal.h"

namespace {

constexpr int MAXV = 800;   // >= 778 vertices
constexpr int NCHUNK = 4;   // workgroups per hand (vertex ranges)
__constant__ int c_parents[16] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14};
__constant__ int c_tips[5] = {744, 320, 443, 554, 671};                       // mano_wrapper.py:23
__constant__ int c_joint_map[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

__global__ __launch_bounds__(256) void mano_kernel(hm_mano_model mm, const float* __restrict__ pose6d,
                                                   const float* __restrict__ betas, const float* __restrict__ cam,
                                                   float* __restrict__ rotmats, float* __restrict__ verts,
                                                   float* __restrict__ joints, float* __restrict__ cam_t,
                                                   float* __restrict__ kp2d, float focal, float image_size) {
  __shared__ float vs[MAXV * 3];     // v_shaped
  __shared__ float vp[MAXV * 3];     // v_posed (this workgroup's vertex range only)
  __shared__ float Rm[16 * 9];
  __shared__ float Jr[16 * 3];
  __shared__ float pf[136];
  __shared__ float G[16 * 12];       // global transforms [R | t], row-major 3x4
  __shared__ float A[16 * 12];       // skinning transforms
  __shared__ float src[21 * 3];      // 16 posed joints + 5 finger tips
  __shared__ float bt[10];

  const int b = blockIdx.x / NCHUNK, chunk = blockIdx.x % NCHUNK, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = mm.n_verts;
  const int vper = (V + NCHUNK - 1) / NCHUNK, v0 = chunk * vper, v1 = min(V, v0 + vper);   // this workgroup's vertices

  if (tid < 16) {   // rot6d -> rotation matrix, columns (b1, b2, b1 x b2)
    const float* x = pose6d + (size_t)b * 96 + tid * 6;
    const float a1x = x[0], a1y = x[1], a1z = x[2], a2x = x[3], a2y = x[4], a2z = x[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    float* R = Rm + tid * 9;
    R[0] = b1x; R[1] = b2x; R[2] = b3x;
    R[3] = b1y; R[4] = b2y; R[5] = b3y;
    R[6] = b1z; R[7] = b2z; R[8] = b3z;
    float* ro = rotmats + ((size_t)b * 16 + tid) * 9;
    if (chunk == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) ro[i] = R[i];
    }
  }
  if (tid >= 64 && tid < 74) bt[tid - 64] = betas[(size_t)b * 10 + (tid - 64)];
  __syncthreads();

  // blend shapes: v_shaped = v_template + shapedirs . betas
  for (int o = tid; o < 3 * V; o += 256) {
    const float* sd = mm.shapedirs + (size_t)o * 10;
    float a = 0.f;
#pragma unroll
    for (int l = 0; l < 10; ++l) a = fmaf(bt[l], sd[l], a);
    vs[o] = mm.v_template[o] + a;
  }
  if (tid < 135) {   // pose feature (R[1:] - I)
    const int e = tid % 9;
    pf[tid] = Rm[9 + tid] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
  }
  __syncthreads();

  // joint regression: J = J_regressor . v_shaped (48 outputs, 12 per wave)
  for (int o = wave * 12; o < wave * 12 + 12; ++o) {
    const int j = o / 3, c = o % 3;
    float a = 0.f;
    for (int v = lane; v < V; v += 64) a = fmaf(mm.J_regressor[(size_t)j * V + v], vs[v * 3 + c], a);
    a = wave_sum(a);
    if (lane == 0) Jr[o] = a;
  }
  // pose blend shapes: v_posed = v_shaped + pose_feature . posedirs (three independent chains, 9 loads in flight)
  for (int o = 3 * v0 + tid; o < 3 * v1; o += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float* pd = mm.posedirs + o;
#pragma unroll 5
    for (int p = 0; p < 135; p += 3) {
      a0 = fmaf(pf[p], pd[(size_t)p * 3 * V], a0);
      a1 = fmaf(pf[p + 1], pd[(size_t)(p + 1) * 3 * V], a1);
      a2 = fmaf(pf[p + 2], pd[(size_t)(p + 2) * 3 * V], a2);
    }
    vp[o] = vs[o] + ((a0 + a1) + a2);
  }
  __syncthreads();

  // kinematic chain by levels (parents before children)
  for (int lev = 0; lev < 4; ++lev) {
    if (tid < 16) {
      const int par = c_parents[tid];
      const int mylev = tid == 0 ? 0 : ((tid - 1) % 3) + 1;
      if (mylev == lev) {
        const float* R = Rm + tid * 9;
        float* g = G + tid * 12;
        if (par < 0) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            g[r * 4 + 0] = R[r * 3 + 0]; g[r * 4 + 1] = R[r * 3 + 1]; g[r * 4 + 2] = R[r * 3 + 2];
            g[r * 4 + 3] = Jr[r];
          }
        } else {
          const float* gp = G + par * 12;
          const float rx = Jr[tid * 3 + 0] - Jr[par * 3 + 0], ry = Jr[tid * 3 + 1] - Jr[par * 3 + 1],
                      rz = Jr[tid * 3 + 2] - Jr[par * 3 + 2];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const float p0 = gp[r * 4 + 0], p1 = gp[r * 4 + 1], p2 = gp[r * 4 + 2];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[r * 4 + c] = p0 * R[0 * 3 + c] + p1 * R[1 * 3 + c] + p2 * R[2 * 3 + c];
            g[r * 4 + 3] = p0 * rx + p1 * ry + p2 * rz + gp[r * 4 + 3];
          }
        }
      }
    }
    __syncthreads();
  }
  if (tid < 16) {   // A = G - pack(G . [J; 0]); posed joint = translation of G
    const float* g = G + tid * 12;
    float* a = A + tid * 12;
    const float jx = Jr[tid * 3 + 0], jy = Jr[tid * 3 + 1], jz = Jr[tid * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      a[r * 4 + 0] = g[r * 4 + 0]; a[r * 4 + 1] = g[r * 4 + 1]; a[r * 4 + 2] = g[r * 4 + 2];
      a[r * 4 + 3] = g[r * 4 + 3] - (g[r * 4 + 0] * jx + g[r * 4 + 1] * jy + g[r * 4 + 2] * jz);
      src[tid * 3 + r] = g[r * 4 + 3];
    }
  }
  __syncthreads();

  // linear blend skinning
  for (int v = v0 + tid; v < v1; v += 256) {
    float Tm[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Tm[e] = 0.f;
    const float* w = mm.lbs_weights + (size_t)v * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float wj = w[j];
#pragma unroll
      for (int e = 0; e < 12; ++e) Tm[e] = fmaf(wj, A[j * 12 + e], Tm[e]);
    }
    const float px = vp[v * 3 + 0], py = vp[v * 3 + 1], pz = vp[v * 3 + 2];
    float o3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) o3[r] = Tm[r * 4 + 0] * px + Tm[r * 4 + 1] * py + Tm[r * 4 + 2] * pz + Tm[r * 4 + 3];
    float* vo = verts + ((size_t)b * V + v) * 3;
    vo[0] = o3[0]; vo[1] = o3[1]; vo[2] = o3[2];
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (v == c_tips[t]) { src[(16 + t) * 3 + 0] = o3[0]; src[(16 + t) * 3 + 1] = o3[1]; src[(16 + t) * 3 + 2] = o3[2]; }
  }
  __syncthreads();

  // joint reorder, camera translation, projection: the 16 chain joints by the hand's first workgroup, a finger tip by
  // the workgroup that skinned its vertex
  const int s_ = tid < 21 ? c_joint_map[tid] : 0;
  if (tid < 21 && (s_ < 16 ? chunk == 0 : (c_tips[s_ - 16] >= v0 && c_tips[s_ - 16] < v1))) {
    const int s = s_;
    const float jx = src[s * 3 + 0], jy = src[s * 3 + 1], jz = src[s * 3 + 2];
    float* jo = joints + ((size_t)b * 21 + tid) * 3;
    jo[0] = jx; jo[1] = jy; jo[2] = jz;
    const float c0 = cam[(size_t)b * 3 + 0], c1 = cam[(size_t)b * 3 + 1], c2 = cam[(size_t)b * 3 + 2];
    const float tz = 2.0f * focal / (image_size * c0 + 1e-9f);
    if (tid == 0 && chunk == 0) { cam_t[(size_t)b * 3 + 0] = c1; cam_t[(size_t)b * 3 + 1] = c2; cam_t[(size_t)b * 3 + 2] = tz; }
    const float px = jx + c1, py = jy + c2, pz = jz + tz;
    const float f = focal / image_size;
    kp2d[((size_t)b * 21 + tid) * 2 + 0] = (px / pz) * f;
    kp2d[((size_t)b * 21 + tid) * 2 + 1] = (py / pz) * f;
  }
}

}  // namespace

extern "C" int hm_mano_forward(const hm_mano_model* model, const float* pose6d, const float* betas, const float* cam,
                               float* rotmats, float* verts, float* joints, float* cam_t, float* kp2d, int B,
                               float focal_length, float image_size, void* stream_) {
  if (!model || !pose6d || !betas || !cam || !rotmats || !verts || !joints || !cam_t || !kp2d || B <= 0)
    return hm_set_error(HM_ERR_ARG, "hm_mano_forward: null pointer or empty batch");
  if (model->n_verts <= 744 || model->n_verts > MAXV)
    return hm_set_error(HM_ERR_ARG, "hm_mano_forward: n_verts must be in (744, 800] (MANO has 778)");
  if (!model->v_template || !model->shapedirs || !model->posedirs || !model->J_regressor || !model->lbs_weights)
    return hm_set_error(HM_ERR_ARG, "hm_mano_forward: incomplete model");
  HmProfScope prof(HM_K_MANO, 0, B, model->n_verts, 0, (hipStream_t)stream_);
  hipLaunchKernelGGL(mano_kernel, dim3(B * NCHUNK), dim3(256), 0, (hipStream_t)stream_, *model, pose6d, betas, cam, rotmats, verts,
                     joints, cam_t, kp2d, focal_length, image_size);
  return hm_check_launch("hm_mano_forward");
}
