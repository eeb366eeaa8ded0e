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

// Mixed-fleet bucketing = a STABLE partition of the frame indices by model id, in three launches and without a single
// contended atomic (the first version took its list slots from one cursor word per model: 8 192 returning atomics on 4
// addresses = 95 us for 131 072 frames, the rate one word sustains; this one is launch-bound, ~5 us per kernel, and
// its index lists are deterministic and keep neighbouring frames together):
//   1. every block histograms ITS contiguous chunk of the batch in LDS -> blockcnt[block][model];
//   2. one block turns the per-block counts into bucket offsets and per-(block, model) list positions (exclusive scan);
//   3. every block walks its chunk again, 256 frames at a time, and writes each frame's index to its model's list at
//      (block base + frames of that model seen so far in the chunk): ballot ranks within a wave, LDS across the waves.
constexpr int FLEET_MAX_BLOCKS = 1024;

__global__ void __launch_bounds__(256) fleet_count_kernel(const int32_t* __restrict__ model_id, int64_t B, int64_t chunk,
                                                          int n_models, int32_t* __restrict__ blockcnt,
                                                          int32_t* __restrict__ bad) {
  __shared__ int32_t h[FLEET_MAX_MODELS + 1];
  if (threadIdx.x <= FLEET_MAX_MODELS) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = b0 + chunk < B ? b0 + chunk : B;
  for (int64_t b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
    const int m = model_id[b];
    atomicAdd(&h[(m >= 0 && m < n_models) ? m : FLEET_MAX_MODELS], 1);  // LDS atomics; unknown ids are counted apart
  }
  __syncthreads();
  if (threadIdx.x < FLEET_MAX_MODELS) blockcnt[(size_t)blockIdx.x * FLEET_MAX_MODELS + threadIdx.x] = h[threadIdx.x];
  if (threadIdx.x == 0 && h[FLEET_MAX_MODELS]) atomicAdd(bad, h[FLEET_MAX_MODELS]);  // frames left untouched
}

// bucket[2m] = first slot of model m in the index list, bucket[2m+1] = its frame count;
// blockcnt[block][m] becomes the list position of block `block`'s first frame of model m.
__global__ void __launch_bounds__(256) fleet_offsets_kernel(int n_models, int nblocks, int32_t* __restrict__ blockcnt,
                                                            int32_t* __restrict__ counts, int32_t* __restrict__ bucket) {
  __shared__ int32_t part[16][FLEET_MAX_MODELS];
  __shared__ int32_t base[FLEET_MAX_MODELS];
  const int m = threadIdx.x & 15, g = threadIdx.x >> 4;  // 16 groups of consecutive blocks x 16 models
  const int per = (nblocks + 15) / 16;
  const int k0 = g * per, k1 = (k0 + per < nblocks) ? k0 + per : nblocks;
  int32_t s = 0;
  for (int k = k0; k < k1; ++k) s += blockcnt[(size_t)k * FLEET_MAX_MODELS + m];
  part[g][m] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t off = 0;
    for (int mm = 0; mm < FLEET_MAX_MODELS; ++mm) {
      int32_t tot = 0;
      for (int gg = 0; gg < 16; ++gg) tot += part[gg][mm];
      base[mm] = off;
      if (mm < n_models) {
        bucket[2 * mm] = off;
        bucket[2 * mm + 1] = tot;
        counts[mm] = tot;
      }
      off += tot;
    }
  }
  __syncthreads();
  int32_t run = base[m];
  for (int gg = 0; gg < g; ++gg) run += part[gg][m];
  for (int k = k0; k < k1; ++k) {
    const int32_t c = blockcnt[(size_t)k * FLEET_MAX_MODELS + m];
    blockcnt[(size_t)k * FLEET_MAX_MODELS + m] = run;
    run += c;
  }
}

__global__ void __launch_bounds__(256) fleet_scatter_kernel(const int32_t* __restrict__ model_id, int64_t B, int64_t chunk,
                                                            int n_models, const int32_t* __restrict__ blockcnt,
                                                            int32_t* __restrict__ perm) {
  __shared__ int32_t run[FLEET_MAX_MODELS];
  __shared__ int32_t wcnt[4][FLEET_MAX_MODELS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < FLEET_MAX_MODELS) run[threadIdx.x] = blockcnt[(size_t)blockIdx.x * FLEET_MAX_MODELS + threadIdx.x];
  const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = b0 + chunk < B ? b0 + chunk : B;
  for (int64_t s0 = b0; s0 < b1; s0 += 256) {  // block-uniform trip count
    const int64_t b = s0 + threadIdx.x;
    const int m = b < b1 ? model_id[b] : -1;
    const bool ok = m >= 0 && m < n_models;
    int rank = 0;
    for (int mm = 0; mm < n_models; ++mm) {  // wave-uniform loop
      const unsigned long long mask = __ballot(ok && m == mm);
      if (lane == 0) wcnt[w][mm] = (int32_t)__popcll(mask);
      if (ok && m == mm) rank = (int)__popcll(mask & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (ok) {
      int32_t pos = run[m] + rank;
      for (int ww = 0; ww < w; ++ww) pos += wcnt[ww][m];
      perm[pos] = (int32_t)b;
    }
    __syncthreads();
    if (threadIdx.x < n_models) run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    __syncthreads();
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

// Hard-frames-first keys of a DexPilot batch WITHOUT a screening launch (dexr_api.hip: launch_wide).  What makes a DexPilot
// frame slow is the pinch projection (/root/reference/src/dex_retargeting/optimizer.py:462-508): a pair vector that is
// projected carries a 200-400 x weight and a target of fixed length, and the frame in which a projection switches on or off
// jumps to a different objective.  Measured on the tracking workload (tools/probe_pred.py, 65 536 frames): "a projection bit
// changed in this frame" flags 3 % of the frames and holds 94 % of those that need >= 24 solver passes (87 % of >= 16); "any
// projection active" flags 14 % and holds 99.8 % (94 %).  Both follow from the keypoints and the incoming state alone --
// the pre-amble restated on one thread per frame, ~20 loads, no kinematics.  key 0: a bit changed; 1: some projection
// active; 2: neither.
struct DexKeyMap {
  int32_t h_task[16], h_origin[16];  // keypoint indices of the first 16 reference rows (-1: the row is kp[h_task])
};
__global__ void __launch_bounds__(256) dexpilot_key_kernel(const float* __restrict__ kpts, const float* __restrict__ ref,
                                                           const uint32_t* __restrict__ state, int64_t B, int n_kp, int n_ref,
                                                           DexKeyMap map, int F, float project_dist, float escape_dist,
                                                           int32_t* __restrict__ key) {
  const int len_s1 = F - 1;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
    auto dist_of = [&](int row) -> float {
      float v[3];
      if (kpts) {
        const float* a = kpts + (b * n_kp + map.h_task[row]) * 3;
        const int o = map.h_origin[row];
        for (int i = 0; i < 3; ++i) v[i] = o >= 0 ? a[i] - kpts[(b * n_kp + o) * 3 + i] : a[i];
      } else {
        for (int i = 0; i < 3; ++i) v[i] = ref[(b * n_ref + row) * 3 + i];
      }
      return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    };
    const uint32_t st = state ? state[b] : 0u;
    uint32_t nst = 0;
    for (int i = 0; i < len_s1; ++i) {
      const float d = dist_of(i);
      bool on = (st >> i) & 1u;
      if (d < project_dist) on = true;
      if (d > escape_dist) on = false;
      nst |= (on ? 1u : 0u) << i;
    }
    int idx = len_s1;
    for (int a = 0; a < F - 2; ++a)
      for (int b2 = a + 1; b2 < F - 1; ++b2) {
        const bool on = ((nst >> b2) & 1u) && ((nst >> a) & 1u) && (dist_of(idx) <= 0.03f);
        nst |= (on ? 1u : 0u) << idx;
        ++idx;
      }
    key[b] = nst != st ? 0 : (nst != 0u ? 1 : 2);
  }
}

// The same keys for ONE MODEL'S BUCKET of a fleet batch (positions [0, count) of its index-list segment; offset and count
// are read from device memory, entries beyond the count get the "not a model" id the bucketing kernels skip), and the gather
// that turns the partition of the POSITIONS into the model's hard-frames-first index list.
__global__ void __launch_bounds__(256) fleet_segment_key_kernel(const float* __restrict__ kpts, const uint32_t* __restrict__ state,
                                                                const int32_t* __restrict__ perm, const int32_t* __restrict__ seg,
                                                                int64_t B, int n_kp, DexKeyMap map, int F, float project_dist,
                                                                float escape_dist, int32_t* __restrict__ key) {
  const int64_t off = seg[0], cnt = seg[1];
  const int len_s1 = F - 1;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < B; p += (int64_t)gridDim.x * blockDim.x) {
    if (p >= cnt) {
      key[p] = -1;
      continue;
    }
    const int64_t b = perm[off + p];
    auto dist_of = [&](int row) -> float {
      const float* a = kpts + (b * n_kp + map.h_task[row]) * 3;
      const int o = map.h_origin[row];
      float v[3];
      for (int i = 0; i < 3; ++i) v[i] = o >= 0 ? a[i] - kpts[(b * n_kp + o) * 3 + i] : a[i];
      return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    };
    const uint32_t st = state[b];
    uint32_t nst = 0;
    for (int i = 0; i < len_s1; ++i) {
      const float d = dist_of(i);
      bool on = (st >> i) & 1u;
      if (d < project_dist) on = true;
      if (d > escape_dist) on = false;
      nst |= (on ? 1u : 0u) << i;
    }
    int idx = len_s1;
    for (int a = 0; a < F - 2; ++a)
      for (int b2 = a + 1; b2 < F - 1; ++b2) {
        const bool on = ((nst >> b2) & 1u) && ((nst >> a) & 1u) && (dist_of(idx) <= 0.03f);
        nst |= (on ? 1u : 0u) << idx;
        ++idx;
      }
    key[p] = nst != st ? 0 : (nst != 0u ? 1 : 2);
  }
}

__global__ void __launch_bounds__(256) fleet_segment_gather_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ seg,
                                                                   const int32_t* __restrict__ pos_sorted, int64_t B,
                                                                   int32_t* __restrict__ out_perm, int32_t* __restrict__ out_seg) {
  const int64_t off = seg[0], cnt = seg[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B && i < cnt; i += (int64_t)gridDim.x * blockDim.x)
    out_perm[i] = perm[off + pos_sorted[i]];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_seg[0] = 0;
    out_seg[1] = (int32_t)cnt;
  }
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
// workspace layout (int32): counts[MAX] | (unused)[MAX] | bucket[2 MAX] | bad[1] | pad | blockcnt[MAX_BLOCKS x MAX] | perm[B]
size_t dexr_fleet_ws_ints() { return 4 * (size_t)FLEET_MAX_MODELS + 16 + (size_t)FLEET_MAX_BLOCKS * FLEET_MAX_MODELS; }

hipError_t dexr_fleet_bucket_launch(int n_models, int64_t B, const int32_t* model_id, int32_t* ws, hipStream_t st) {
  int32_t* counts = ws;
  int32_t* bucket = ws + 2 * FLEET_MAX_MODELS;
  int32_t* bad = ws + 4 * FLEET_MAX_MODELS;
  int32_t* blockcnt = ws + 4 * FLEET_MAX_MODELS + 16;
  int32_t* perm = ws + dexr_fleet_ws_ints();
  hipError_t e = hipMemsetAsync(ws, 0, (4 * (size_t)FLEET_MAX_MODELS + 16) * sizeof(int32_t), st);
  if (e != hipSuccess) return e;
  // contiguous chunks of >= 2 048 frames, at most FLEET_MAX_BLOCKS of them
  int64_t chunk = (B + FLEET_MAX_BLOCKS - 1) / FLEET_MAX_BLOCKS;
  chunk = (chunk + 255) / 256 * 256;
  if (chunk < 2048) chunk = 2048;
  const int nblocks = (int)((B + chunk - 1) / chunk);
  hipLaunchKernelGGL(fleet_count_kernel, dim3(nblocks), dim3(256), 0, st, model_id, B, chunk, n_models, blockcnt, bad);
  hipLaunchKernelGGL(fleet_offsets_kernel, dim3(1), dim3(256), 0, st, n_models, nblocks, blockcnt, counts, bucket);
  hipLaunchKernelGGL(fleet_scatter_kernel, dim3(nblocks), dim3(256), 0, st, model_id, B, chunk, n_models, blockcnt, perm);
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

// DexPilot batches: keys from the projection state (dexpilot_key_kernel), then the index list through the bucketing kernels
hipError_t dexr_dexpilot_order_launch(int64_t B, const float* kpts, const float* ref, const uint32_t* state, int n_kp, int n_ref,
                                      const int32_t* h_task, const int32_t* h_origin, int F, float project_dist, float escape_dist,
                                      int32_t* key, int32_t* ws, hipStream_t st) {
  DexKeyMap map;
  for (int i = 0; i < 16; ++i) {
    map.h_task[i] = h_task[i];
    map.h_origin[i] = h_origin[i];
  }
  const int64_t want = (B + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(dexpilot_key_kernel, dim3(blocks), dim3(256), 0, st, kpts, ref, state, B, n_kp, n_ref, map, F, project_dist,
                     escape_dist, key);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return dexr_fleet_bucket_launch(3, B, key, ws, st);
}

// TAIL LIST of a large batch (dexr_api.hip: launch_wide): the frames the first launch left at its pass cap (status MAXITER) become
// the index list of the second, one-frame-per-wave launch; their status / final-value entries are cleared for it to set.
namespace {
__global__ void __launch_bounds__(256) tail_key_kernel(int32_t* __restrict__ status, float* __restrict__ fval, int64_t B,
                                                       int32_t* __restrict__ key) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
    const bool tail = status[b] == 1;  // DEXR_STATUS_MAXITER
    key[b] = tail ? 0 : -1;
    if (tail) {
      status[b] = 0;
      if (fval) fval[b] = 0.f;
    }
  }
}
}  // namespace
hipError_t dexr_tail_list_launch(int64_t B, int32_t* status, float* fval, int32_t* key, int32_t* ws, hipStream_t st) {
  const int64_t want = (B + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(tail_key_kernel, dim3(blocks), dim3(256), 0, st, status, fval, B, key);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return dexr_fleet_bucket_launch(1, B, key, ws, st);  // (key -1: skipped; bucket 0 = the tail, stable order)
}

// One DexPilot model's bucket of a fleet batch, hard frames first: extra workspace (int32) = key[B] | bucketing workspace
// (dexr_fleet_ws_ints() + B) | out_perm[B] | out_seg[2] -- dexr_fleet_order_ws_ints(B) in all
size_t dexr_fleet_order_ws_ints(int64_t B) { return 3 * (size_t)(B > 0 ? B : 0) + dexr_fleet_ws_ints() + 4; }

hipError_t dexr_fleet_dexpilot_order_launch(int64_t B, const float* kpts, const uint32_t* state, const int32_t* perm, const int32_t* seg,
                                            int n_kp, const int32_t* h_task, const int32_t* h_origin, int F, float project_dist,
                                            float escape_dist, int32_t* xws, const int32_t** out_perm, const int32_t** out_seg,
                                            hipStream_t st) {
  DexKeyMap map;
  for (int i = 0; i < 16; ++i) {
    map.h_task[i] = h_task[i];
    map.h_origin[i] = h_origin[i];
  }
  int32_t* key = xws;
  int32_t* ws2 = key + B;
  int32_t* operm = ws2 + dexr_fleet_ws_ints() + B;
  int32_t* oseg = operm + B;
  const int64_t want = (B + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(fleet_segment_key_kernel, dim3(blocks), dim3(256), 0, st, kpts, state, perm, seg, B, n_kp, map, F, project_dist,
                     escape_dist, key);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = dexr_fleet_bucket_launch(3, B, key, ws2, st);  // (positions beyond the bucket carry id -1: skipped)
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fleet_segment_gather_kernel, dim3(blocks), dim3(256), 0, st, perm, seg, ws2 + dexr_fleet_ws_ints(), B, operm, oseg);
  *out_perm = operm;
  *out_seg = oseg;
  return hipGetLastError();
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
