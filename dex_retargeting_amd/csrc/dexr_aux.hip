// dexr_aux.hip -- the small HBM-bound kernels around the solve:
//   * mixed-fleet bucketing (BASELINE.json configs[4]): frames -> per-model index lists, entirely on the device;
//   * per-frame bookkeeping of SeqRetargeting for T x B frames: robot-qpos composition, mimic fill, low-pass filter
//     (/root/reference/src/dex_retargeting/seq_retarget.py:125-133, kinematics_adaptor.py:102-105,
//     optimizer_utils.py:7-13).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dexr.h"

namespace {

constexpr int FLEET_MAX_MODELS = DEXR_FLEET_MAX_MODELS;

// counts[m] = number of frames of model m.  One LDS histogram per block, then one atomic per model and block.
__global__ void __launch_bounds__(256) fleet_count_kernel(const int32_t* __restrict__ model_id, int64_t B, int n_models,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ bad) {
  __shared__ int32_t h[FLEET_MAX_MODELS];
  if (threadIdx.x < FLEET_MAX_MODELS) h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
    const int m = model_id[b];
    if (m >= 0 && m < n_models) atomicAdd(&h[m], 1);
    else atomicAdd(bad, 1);  // frames with an unknown model id are left untouched and counted
  }
  __syncthreads();
  if (threadIdx.x < n_models && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}

// bucket[2m] = first slot of model m in the index list, bucket[2m+1] = its frame count; cursors reset.
__global__ void fleet_offsets_kernel(int n_models, const int32_t* __restrict__ counts, int32_t* __restrict__ bucket,
                                     int32_t* __restrict__ cursor) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int32_t off = 0;
    for (int m = 0; m < n_models; ++m) {
      bucket[2 * m] = off;
      bucket[2 * m + 1] = counts[m];
      cursor[m] = 0;
      off += counts[m];
    }
  }
}

// perm[bucket[2m] + k] = b for the k-th frame b of model m.  Within a wave, lanes of the same model take consecutive
// slots from ONE atomic (match-any by ballot over the model id), so the list keeps runs of neighbouring frames together.
__global__ void __launch_bounds__(256) fleet_scatter_kernel(const int32_t* __restrict__ model_id, int64_t B, int n_models,
                                                            const int32_t* __restrict__ bucket, int32_t* __restrict__ cursor,
                                                            int32_t* __restrict__ perm) {
  const int lane = threadIdx.x & 63;
  for (int64_t b0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane); b0 < B; b0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = b0 + lane;
    const int m = b < B ? model_id[b] : -1;
    const bool ok = m >= 0 && m < n_models;
    for (int mm = 0; mm < n_models; ++mm) {  // wave-uniform loop: one atomic per (wave, model present in it)
      const unsigned long long mask = __ballot(ok && m == mm);
      if (mask == 0ull) continue;
      const int leader = __ffsll((long long)mask) - 1;
      int32_t base = 0;
      if (lane == leader) base = atomicAdd(&cursor[mm], (int32_t)__popcll(mask));
      base = __shfl(base, leader, 64);
      if (ok && m == mm) {
        const int rank = (int)__popcll(mask & ((1ull << lane) - 1ull));
        perm[bucket[2 * mm] + base + rank] = (int32_t)b;
      }
    }
  }
}

// Longest-first ordering of a batch (dexr_api.hip: launch_wide): key 0 = frame whose objective at the start point is
// above `ratio` x the batch mean (it will need many solver passes), key 1 = everything else.
__global__ void __launch_bounds__(256) lpt_key_kernel(const float* __restrict__ f0, const float* __restrict__ sum, int64_t B,
                                                      float ratio, int32_t* __restrict__ key) {
  const float thr = ratio * (*sum) / (float)B;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x)
    key[b] = f0[b] > thr ? 0 : 1;
}

struct ComposeMap {
  int32_t kind[DEXR_MAX_DOF];  // 0 target joint, 1 fixed joint, 2 mimic joint
  int32_t idx[DEXR_MAX_DOF];   // column of qpos_raw / column of fixed / source dof
  double mult[DEXR_MAX_DOF], off[DEXR_MAX_DOF];
};

// One thread per (sequence, dof): walks the T frames in order carrying the filter output y.
__global__ void __launch_bounds__(256) seq_compose_kernel(int64_t B, int T, int n_q, int n_opt, int n_fixed, ComposeMap map,
                                                          const float* __restrict__ qraw, const float* __restrict__ fixed,
                                                          double alpha, int use_filter, int first_frame_initialises,
                                                          double* __restrict__ filt, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n_q) return;
  const int64_t b = i / n_q;
  const int j = (int)(i - b * n_q);
  // the value of dof j comes from: a target joint (the optimiser's float32 answer), a fixed joint (caller input) or
  // a mimic joint = source * multiplier + offset, evaluated in float64 on the composed vector like
  // MimicJointKinematicAdaptor.forward_qpos (kinematics_adaptor.py:102-105)
  int kind = map.kind[j], idx = map.idx[j];
  double mult = 1.0, off = 0.0;
  if (kind == 2) {
    mult = map.mult[j];
    off = map.off[j];
    const int s = idx;
    kind = map.kind[s];
    idx = map.idx[s];
  }
  double y = use_filter ? filt[i] : 0.0;
  for (int t = 0; t < T; ++t) {
    const int64_t row = (int64_t)t * B + b;
    double v = 0.0;  // robot_qpos starts as zeros (seq_retarget.py:125)
    if (kind == 0) v = (double)qraw[row * n_opt + idx];
    else if (kind == 1 && fixed) v = (double)fixed[row * n_fixed + idx];
    v = v * mult + off;
    if (use_filter) {  // LPFilter.next (optimizer_utils.py:7-13)
      if (t == 0 && first_frame_initialises) y = v;
      else y = y + alpha * (v - y);
      v = y;
    }
    out[row * n_q + j] = v;
  }
  if (use_filter) filt[i] = y;
}

}  // namespace

// ---- launch helpers used by dexr_api.hip --------------------------------------------------------------------------
// workspace layout (int32): counts[MAX] | cursor[MAX] | bucket[2 MAX] | bad[1] | pad | perm[B]
size_t dexr_fleet_ws_ints() { return 4 * (size_t)FLEET_MAX_MODELS + 16; }

hipError_t dexr_fleet_bucket_launch(int n_models, int64_t B, const int32_t* model_id, int32_t* ws, hipStream_t st) {
  int32_t* counts = ws;
  int32_t* cursor = ws + FLEET_MAX_MODELS;
  int32_t* bucket = ws + 2 * FLEET_MAX_MODELS;
  int32_t* bad = ws + 4 * FLEET_MAX_MODELS;
  int32_t* perm = ws + dexr_fleet_ws_ints();
  hipError_t e = hipMemsetAsync(ws, 0, dexr_fleet_ws_ints() * sizeof(int32_t), st);
  if (e != hipSuccess) return e;
  const int64_t want = (B + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(fleet_count_kernel, dim3(blocks), dim3(256), 0, st, model_id, B, n_models, counts, bad);
  hipLaunchKernelGGL(fleet_offsets_kernel, dim3(1), dim3(64), 0, st, n_models, counts, bucket, cursor);
  hipLaunchKernelGGL(fleet_scatter_kernel, dim3(blocks), dim3(256), 0, st, model_id, B, n_models, bucket, cursor, perm);
  return hipGetLastError();
}

// keys from the screening launch's F(x0) values, then the index list (hard frames first) through the fleet bucketing
// kernels; ws: the fleet workspace (perm at ws + dexr_fleet_ws_ints())
hipError_t dexr_lpt_order_launch(int64_t B, const float* f0, const float* sum, float ratio, int32_t* key, int32_t* ws, hipStream_t st) {
  const int64_t want = (B + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(lpt_key_kernel, dim3(blocks), dim3(256), 0, st, f0, sum, B, ratio, key);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return dexr_fleet_bucket_launch(2, B, key, ws, st);
}

hipError_t dexr_seq_compose_launch(int64_t B, int T, int n_q, int n_opt, int n_fixed, const int32_t* kind,
                                   const int32_t* idx, const double* mult, const double* off, const float* qraw,
                                   const float* fixed, double alpha, int use_filter, int first_frame_initialises,
                                   double* filt, double* out, hipStream_t st) {
  ComposeMap map;
  for (int j = 0; j < DEXR_MAX_DOF; ++j) {
    map.kind[j] = j < n_q ? kind[j] : 0;
    map.idx[j] = j < n_q ? idx[j] : 0;
    map.mult[j] = j < n_q ? mult[j] : 1.0;
    map.off[j] = j < n_q ? off[j] : 0.0;
  }
  const int64_t n = B * n_q;
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(seq_compose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, B, T, n_q, n_opt, n_fixed, map, qraw,
                     fixed, alpha, use_filter, first_frame_initialises, filt, out);
  return hipGetLastError();
}
