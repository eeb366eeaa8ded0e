// dexr_prep.hip -- the step right before the hot path: raw detector keypoints ->
// wrist-centred keypoints in the MANO frame, for B frames at once.
//
//   kp_c      = kp - kp[0]                                           (single_hand_detector.py:102)
//   R         = estimate_frame_from_hand_points(kp_c)                (single_hand_detector.py:129-158)
//   joint_pos = kp_c @ R @ operator2mano                             (single_hand_detector.py:104; constants.py:7-21)
//
// The reference fits the palm plane through keypoints (0, 5, 9) with an SVD of the three centred points and takes
// the least-significant right singular vector as the normal.  Three points always span (at most) a plane, so that
// vector is +-normalize((p5-p0) x (p9-p0)); the sign LAPACK happens to return does not matter because the
// reference's own final orientation test (`z . (p5 - p9) < 0` -> flip normal and z) fixes it.  x never depends on it.
//
// HBM-bound byte shuffling: 252 B in + 252 B out (+ 36 B for R when asked for) per frame.  A workgroup stages a
// tile of 64 frames (16 128 B, contiguous) through LDS with 16-byte loads, 64 lanes derive the per-frame 3x3 matrix
// M = R @ operator2mano in float64, then all 256 lanes stream the rotated keypoints back with coalesced stores.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dexr.h"

namespace dexr {

constexpr int PREP_NKP = 21;
constexpr int PREP_FPF = PREP_NKP * 3;  // floats per frame
constexpr int PREP_TILE = 64;           // frames per workgroup
constexpr int PREP_THREADS = 256;

struct Mat3 {
  float v[9];
};

__global__ __launch_bounds__(PREP_THREADS) void mano_keypoints_kernel(const float* __restrict__ kp,
                                                                      float* __restrict__ out,
                                                                      float* __restrict__ rot, int64_t B, Mat3 op) {
  __shared__ float s[PREP_TILE * PREP_FPF];
  __shared__ float sm[PREP_TILE * 9];
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * PREP_TILE;
  const int nf = (int)((B - f0) < PREP_TILE ? (B - f0) : PREP_TILE);
  const int n = nf * PREP_FPF;
  const float* src = kp + f0 * PREP_FPF;
  float* dst = out + f0 * PREP_FPF;
  // tile base = f0 * 252 B = blockIdx * 16128 B: 16-byte aligned whenever the tensor is
  const bool vec_in = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  const int n4 = vec_in ? (n >> 2) : 0;
  for (int i = tid; i < n4; i += PREP_THREADS) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    reinterpret_cast<float4*>(s)[i] = v;
  }
  for (int i = (n4 << 2) + tid; i < n; i += PREP_THREADS) s[i] = src[i];
  __syncthreads();

  if (tid < nf) {
    const float* p = s + tid * PREP_FPF;
    double a[3], b[3], nrm[3], x[3], z[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      a[i] = (double)p[5 * 3 + i] - (double)p[i];  // p5 - p0
      b[i] = (double)p[9 * 3 + i] - (double)p[i];  // p9 - p0
    }
    nrm[0] = a[1] * b[2] - a[2] * b[1];
    nrm[1] = a[2] * b[0] - a[0] * b[2];
    nrm[2] = a[0] * b[1] - a[1] * b[0];
    double inv = 1.0 / sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) nrm[i] *= inv;
    // x_vector = points[0] - points[2] = -(p9 - p0); Gram-Schmidt against the normal
    double xd = -(b[0] * nrm[0] + b[1] * nrm[1] + b[2] * nrm[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = -b[i] - xd * nrm[i];
    inv = 1.0 / sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] *= inv;
    z[0] = x[1] * nrm[2] - x[2] * nrm[1];
    z[1] = x[2] * nrm[0] - x[0] * nrm[2];
    z[2] = x[0] * nrm[1] - x[1] * nrm[0];
    // "the vector from pinky to index is similar to the z axis in MANO convention": points[1] - points[2] = p5 - p9
    const double sgn = (z[0] * (a[0] - b[0]) + z[1] * (a[1] - b[1]) + z[2] * (a[2] - b[2])) < 0.0 ? -1.0 : 1.0;
    double R[9];  // frame = stack([x, normal, z], axis=1): columns
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      R[i * 3 + 0] = x[i];
      R[i * 3 + 1] = sgn * nrm[i];
      R[i * 3 + 2] = sgn * z[i];
    }
    if (rot) {
      float* r = rot + (f0 + tid) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) r[i] = (float)R[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        sm[tid * 9 + i * 3 + j] =
            (float)(R[i * 3 + 0] * (double)op.v[0 * 3 + j] + R[i * 3 + 1] * (double)op.v[1 * 3 + j] +
                    R[i * 3 + 2] * (double)op.v[2 * 3 + j]);
  }
  __syncthreads();

  for (int e = tid; e < n; e += PREP_THREADS) {
    const int f = e / PREP_FPF;
    const int r = e - f * PREP_FPF;
    const int pt = r / 3;
    const int c = r - pt * 3;
    const float* p = s + f * PREP_FPF;
    const float* m = sm + f * 9;
    const float* q = p + pt * 3;
    dst[e] = (q[0] - p[0]) * m[c] + (q[1] - p[1]) * m[3 + c] + (q[2] - p[2]) * m[6 + c];
  }
}

}  // namespace dexr

int dexr_prep_launch(int64_t B, const float* kp, const float* op9, float* out, float* rot, hipStream_t st) {
  dexr::Mat3 op;
  for (int i = 0; i < 9; ++i) op.v[i] = op9[i];
  const int64_t blocks = (B + dexr::PREP_TILE - 1) / dexr::PREP_TILE;
  hipLaunchKernelGGL(dexr::mano_keypoints_kernel, dim3((unsigned)blocks), dim3(dexr::PREP_THREADS), 0, st, kp, out, rot,
                     B, op);
  return (int)hipGetLastError();
}
