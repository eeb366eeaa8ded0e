// dexr_api.hip -- C ABI of libdexr.so (see include/dexr.h): table loading, launch geometry, host-pointer
// convenience wrappers.  No torch, no exceptions across the boundary.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "dexr.h"
#include "dexr_gen.hpp"
#include "dexr_hostctx.hpp"
#include "dexr_launch.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(DEXR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
  } while (0)

}  // namespace

// error reporting for the other translation units of the library (dexr_comm.hip)
int dexr_set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

struct dexr_model {
  dexr_model_header h;
  std::vector<dexr_comp_table> comps;
  dexr_comp_table* d_comps = nullptr;
  int bucket = 0;      // NMAX instantiation used for every component of this model
  int lds_frames = 1;  // max n_frame over components
  int lds_terms = 1;   // max n_term over components
  bool chain = false;  // every component is a plain serial chain filling its bucket (CHAIN kernel applies)
  bool tip = false;    // ... and carries one vector term from a base frame to a frame on its last joint (dexr_tip.hpp)
  bool quad = false;   // dense 9..24-joint components solved four lanes per frame (dexr_quad_kernel)
  bool big = false;    // components of 9+ joints solved by dexr_big_kernel (Hessian in LDS, float64 kinematics)
  bool wide = false;   // dense 9..32-joint components solved sixteen lanes per frame (dexr_wide_kernel)
  bool wide_ok = false;  // the model fits that kernel (<= 16 root-to-leaf chains of <= 16 joints; with mimic joints:
                         // <= 16 variables moving <= 3 joints each)
  bool wide_mimic = false;  // ... through its variable-grid instantiation (mimic joints folded)
  int wbucket = 0;          // grid size of the sixteen-lane kernel that serves the model (16 for components of 5-16 joints)
  // longest-first ordering (launch_wide): LSLOTS workspaces handed out round-robin; a slot's last use is fenced by an
  // event, so launches on different streams never share one
  static constexpr int LSLOTS = 4;
  struct LptSlot {
    void* buf = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
  };
  mutable LptSlot lpt[LSLOTS];
  mutable std::atomic<unsigned> lnext{0};
  bool wide_modchol = false;  // ... with modified Cholesky and the damping rules that go with it (tune.pivot_rule)
  std::vector<dexr::WideTable> wide_tabs;
  dexr::WideTable* d_wide = nullptr;
  bool red = false;    // solved in reduced variables by dexr_red_kernel (Hessian of the variables in registers)
  int red_nv = 0;      // its variable bucket (8 or 16), 0 when the model does not fit
  int max_joints = 0, max_vars = 0;
  int big_nh_rows = 0; // n_max (n_max + 1) / 2
  // work-queue heads for the persistent-lane kernels: QSLOTS independent sets of n_comp counters handed out
  // round-robin, so launches in flight on different streams never share a queue
  static constexpr int QSLOTS = 64;
  unsigned* d_queue = nullptr;
  mutable std::atomic<unsigned> qnext{0};
  int n_cu = 256;
  int max_slot = -1;      // deepest saved-transform slot any component uses
  bool has_mimic = false;
  dexr_tuning tune;       // launch / damping parameters (dexr_model_set_tuning); never read from the environment
  bool gen = false;       // generic table (dexr_tables.h): served by the general kernel (dexr_gen.hpp), every mode, float64
  dexr::GenTab gen_tab;   // device pointers into d_gen
  void* d_gen = nullptr;
  float lam_fastdec_user = -1.f;  // >= 0: the caller overrode dexr_tuning.lam_fastdec (same rule as lam_jump below)
  float lam_jump_user = -1.f;  // >= 0: the caller overrode dexr_tuning.lam_jump; otherwise every launch uses the default
                               // of the kernel family it actually dispatches (family_lam_jump)
  mutable dexr::HostCtx host;  // private stream + persistent pinned / device staging of the host-pointer entry points
};

namespace {

int pick_bucket(int nj) {
  const int buckets[] = {4, 8, 16, 24, 32};
  for (int b : buckets)
    if (nj <= b) return b;
  return -1;
}

// Default damping jump of a kernel family.  The parameter's MEANING differs per family: the quad kernel and the plain
// sixteen-lane kernel scale the curvature along the failed step (1.0), the others scale mean diag H (3.0 for the small
// components, 0.3 from 9 joints on).  float64 launches (precision = 1, the polish pass) always run the register kernel.
enum Family { FAM_REGISTER, FAM_QUAD, FAM_BIG, FAM_RED, FAM_WIDE };
float family_lam_jump(const dexr_model* m, Family fam) {
  if (m->lam_jump_user >= 0.f) return m->lam_jump_user;
  if (fam == FAM_QUAD || (fam == FAM_WIDE && !m->wide_modchol)) return 1.0f;
  return m->bucket <= 8 ? 3.0f : 0.3f;
}
// Default of dexr_tuning.lam_fastdec (lambda x this after a step the model predicted to 90 %, instead of x 1/3) per family.
// After a rejection lambda jumps to the curvature scale; at 1/3 per accepted step the sixteen-lane kernel then spent ~10
// over-damped passes coming back.  Measured on it (65 536 frames, tools/all_configs.py with DEXR_TOOL_KNOBS=lam_fastdec=...):
// 0.1 for joint-space models, for the first 6 rejections of a solve (KernelParams::fastdec_max_rej: unbounded it is 5-11 %
// faster still, but a handful of LEAP DexPilot frames then cycle until max_iter) -- same box, Shadow + free joints -7 %,
// LEAP DexPilot -7 %, LEAP / Allegro position -3 %, Shadow vector -4 %, Shadow DexPilot -0.5 %, Allegro DexPilot +4 %;
// models with mimic joints lose 2-6 % with it (Inspire) and keep the 1/3 rule.  The other large-component kernels
// (float64 validation / polish launches run the register kernel) keep 0: with 0.1 a Shadow DexPilot frame of
// tests/test_gpu_parity.py::test_solve_matches_oracle_tracking stops 0.02 rad short of its minimum.
float family_lam_fastdec(const dexr_model* m, Family fam) {
  if (m->lam_fastdec_user >= 0.f) return m->lam_fastdec_user;
  if (fam == FAM_WIDE && !m->has_mimic) return 0.1f;
  return m->bucket <= 8 ? 0.1f : 0.f;
}
Family selected_family(const dexr_model* m) {
  return m->red ? FAM_RED : m->wide ? FAM_WIDE : m->quad ? FAM_QUAD : m->big ? FAM_BIG : FAM_REGISTER;
}

void fill_params(const dexr_model* m, dexr::KernelParams& kp, int64_t B) {
  std::memset(&kp, 0, sizeof(kp));
  const dexr_model_header& h = m->h;
  kp.comps = m->d_comps;
  kp.B = B;
  kp.n_comp = h.n_comp;
  // damping dynamics (measured, tools/term_sweep.py): a rejected step raises lambda at least to lam_jump x the mean
  // curvature instead of creeping up by x2, x4, ...; small components also drop it by 10x (not 3x) after a step the
  // model predicted well.  Allegro vector, 65 536 frames: 0.143 -> 0.119 ms; Shadow DexPilot: 19.4 -> 15.4 ms.
  kp.lam_jump = family_lam_jump(m, FAM_REGISTER);  // launch() sets the dispatched family's value
  kp.lam_fastdec = family_lam_fastdec(m, FAM_REGISTER);  // (likewise)
  kp.lam_recover = m->tune.lam_recover;
  kp.fastdec_max_rej = 6;  // (3 / 6 / 12 / unbounded compared in one box: 6 is the largest bound under which every frame of
                           // the 39 configs still converges within max_iter; profiles/r03_wide_ab_fastdec_same_box.txt)
  kp.floor_scale = m->tune.floor_scale;
  kp.step_cap = m->tune.step_cap;
  kp.blind_tol = 0.f;  // set from the tolerance in apply_options
  kp.n_opt = h.n_opt;
  kp.n_fixed = h.n_fixed;
  kp.n_ref = h.n_ref;
  kp.n_q = h.n_q;
  kp.kind = h.kind;
  kp.num_fingers = h.num_fingers;
  kp.huber_delta = h.huber_delta;
  kp.norm_delta = h.norm_delta;
  kp.scaling = h.scaling;
  kp.inv_norm = h.inv_norm;
  kp.project_dist = h.project_dist;
  kp.escape_dist = h.escape_dist;
  kp.eta1 = h.eta1;
  kp.eta2 = h.eta2;
  kp.lds_frames = m->lds_frames;
  kp.lds_terms = m->lds_terms;
  kp.n_kp = h.n_keypoints;
  kp.ld = h.n_opt;  // plain batches: rows of last / qout are n_opt long
  kp.ldf = h.n_fixed;
  for (int i = 0; i < DEXR_MAXT; ++i) {
    kp.h_origin[i] = h.human_origin[i];
    kp.h_task[i] = h.human_task[i];
  }
}

// launch geometry: one wave per (64-item tile, component); waves of a block sit on consecutive components so
// that the rows of ref/last they share are fetched by one CU.
int launch_big(const dexr_model* m, dexr::KernelParams kp, hipStream_t st) {
  const size_t lds = (size_t)64 * (4 * (size_t)m->big_nh_rows + 8 * 3 * (size_t)m->lds_frames);
  kp.big_nh_rows = m->big_nh_rows;
  // persistent lanes: as many waves as fit a CU's 160 KB of LDS (and one per SIMD) are resident, each starts with a
  // static 64-frame tile, the per-component queue hands out the rest
  const int64_t tiles = (kp.B + 63) / 64;
  int64_t per_cu = (int64_t)((160 * 1024) / (lds > 0 ? lds : 1));
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  int64_t resident = (int64_t)m->n_cu * per_cu;
  if (m->tune.resident_waves > 0) resident = m->tune.resident_waves;
  int64_t per_comp = (resident + kp.n_comp - 1) / kp.n_comp;
  if (per_comp > tiles) per_comp = tiles;
  const int64_t blocks = per_comp * kp.n_comp;
  if (blocks > 0x7fffffffLL) return fail(DEXR_ERR_INVALID, "batch too large for one launch");
  kp.q0 = (uint32_t)(per_comp * 64);
  const unsigned slot = m->qnext.fetch_add(1u) % dexr_model::QSLOTS;
  kp.queue = m->d_queue + (size_t)slot * kp.n_comp;
  hipError_t qe = hipMemsetAsync(kp.queue, 0, (size_t)kp.n_comp * sizeof(unsigned), st);
  if (qe != hipSuccess) return fail(DEXR_ERR_HIP, "queue reset failed: %s", hipGetErrorString(qe));
  dexr::launch_fn fn = dexr::find_big_launcher(m->bucket);
  if (!fn) return fail(DEXR_ERR_UNSUPPORTED, "no large-component kernel for bucket %d", m->bucket);
  hipError_t e = fn(kp, dim3((unsigned)blocks), dim3(64), lds, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return DEXR_OK;
}

int launch_red(const dexr_model* m, dexr::KernelParams kp, hipStream_t st) {
  const size_t lds = (size_t)64 * (4 * 6 * (size_t)m->max_joints + 8 * 3 * (size_t)m->lds_frames);
  kp.red_nj = m->max_joints;
  // persistent lanes as in launch_big: the resident set is what the LDS (160 KB per CU) and the kernel's registers
  // (one wave per SIMD) allow; each wave starts with a static 64-frame tile
  const int64_t tiles = (kp.B + 63) / 64;
  int64_t per_cu = (int64_t)((160 * 1024) / (lds > 0 ? lds : 1));
  const int64_t reg_cap = 4;  // both instantiations are built for one wave per SIMD
  per_cu = per_cu < 1 ? 1 : (per_cu > reg_cap ? reg_cap : per_cu);
  int64_t resident = (int64_t)m->n_cu * per_cu;
  if (m->tune.resident_waves > 0) resident = m->tune.resident_waves;
  int64_t per_comp = (resident + kp.n_comp - 1) / kp.n_comp;
  if (per_comp > tiles) per_comp = tiles;
  const int64_t blocks = per_comp * kp.n_comp;
  if (blocks > 0x7fffffffLL) return fail(DEXR_ERR_INVALID, "batch too large for one launch");
  kp.q0 = (uint32_t)(per_comp * 64);
  const unsigned slot = m->qnext.fetch_add(1u) % dexr_model::QSLOTS;
  kp.queue = m->d_queue + (size_t)slot * kp.n_comp;
  hipError_t qe = hipMemsetAsync(kp.queue, 0, (size_t)kp.n_comp * sizeof(unsigned), st);
  if (qe != hipSuccess) return fail(DEXR_ERR_HIP, "queue reset failed: %s", hipGetErrorString(qe));
  dexr::launch_fn fn = dexr::find_red_launcher(m->red_nv);
  if (!fn) return fail(DEXR_ERR_UNSUPPORTED, "no reduced-variable kernel for %d variables", m->max_vars);
  hipError_t e = fn(kp, dim3((unsigned)blocks), dim3(64), lds, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return DEXR_OK;
}

int launch_quad(const dexr_model* m, dexr::KernelParams kp, hipStream_t st) {
  const size_t per_wave = (size_t)64 * (8 * 3 * (size_t)m->lds_frames + 4 * 3 * (size_t)m->bucket + 4 * (size_t)m->bucket);
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 150 * 1024) wpb >>= 1;
  // persistent quads: at most one wave per SIMD is resident (the kernel needs the whole register file); each starts with
  // a static tile of 16 frames, the rest of the batch is handed out through the per-component queue counter
  const int64_t tiles = (kp.B + 15) / 16;  // 16 frames per wave at a time
  int64_t resident = (int64_t)m->n_cu * 4;
  if (m->tune.resident_waves > 0) resident = m->tune.resident_waves;
  int64_t per_comp = (resident + kp.n_comp - 1) / kp.n_comp;
  if (per_comp > tiles) per_comp = tiles;
  const int64_t waves = per_comp * kp.n_comp;
  const int64_t blocks = (waves + wpb - 1) / wpb;
  if (blocks > 0x7fffffffLL) return fail(DEXR_ERR_INVALID, "batch too large for one launch");
  kp.q0 = (uint32_t)(per_comp * 16);
  const unsigned slot = m->qnext.fetch_add(1u) % dexr_model::QSLOTS;
  kp.queue = m->d_queue + (size_t)slot * kp.n_comp;
  hipError_t qe = hipMemsetAsync(kp.queue, 0, (size_t)kp.n_comp * sizeof(unsigned), st);
  if (qe != hipSuccess) return fail(DEXR_ERR_HIP, "queue reset failed: %s", hipGetErrorString(qe));
  dexr::launch_fn fn = dexr::find_quad_launcher(m->bucket);
  if (!fn) return fail(DEXR_ERR_UNSUPPORTED, "no quad kernel for bucket %d", m->bucket);
  hipError_t e = fn(kp, dim3((unsigned)blocks), dim3(64 * wpb), per_wave * wpb, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return DEXR_OK;
}

}  // namespace
size_t dexr_fleet_ws_ints();
size_t dexr_fleet_order_ws_ints(int64_t B);
hipError_t dexr_fleet_dexpilot_order_launch(int64_t B, const float* kpts, const uint32_t* state, const int32_t* perm, const int32_t* seg,
                                            int n_kp, const int32_t* h_task, const int32_t* h_origin, int F, float project_dist,
                                            float escape_dist, int32_t* xws, const int32_t** out_perm, const int32_t** out_seg,
                                            hipStream_t st);
hipError_t dexr_lpt_order_launch(int64_t B, const float* f0, const float* sum, float ratio, int32_t* key, int32_t* ws, hipStream_t st);
hipError_t dexr_tail_list_launch(int64_t B, int32_t* status, float* fval, int32_t* key, int32_t* ws, hipStream_t st);
hipError_t dexr_dexpilot_order_launch(int64_t B, const float* kpts, const float* ref, const uint32_t* state, int n_kp, int n_ref,
                                      const int32_t* h_task, const int32_t* h_origin, int F, float project_dist, float escape_dist,
                                      int32_t* key, int32_t* ws, hipStream_t st);
namespace {

// sixteen lanes per frame: four frames per wave, two waves per SIMD resident; persistent rows fed like the quads
int launch_wide_once(const dexr_model* m, dexr::KernelParams kp, hipStream_t st, bool tail_launch = false) {
  const size_t per_wave = m->wide_mimic ? dexr::wide_lds_per_wave_m_16() : dexr::wide_lds_per_wave(m->wbucket);
  // (a row copies its frame's input block -- keypoints or ref_value rows -- into the 1 KB term-block area of its LDS slot)
  if (kp.kpts && kp.n_kp > 85) return fail(DEXR_ERR_UNSUPPORTED, "the sixteen-lane kernel takes at most 85 keypoints per frame (%d)", kp.n_kp);
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 80 * 1024) wpb >>= 1;
  // ONE FRAME PER WAVE for small plain batches (dexr_wide.hpp SPRINT; dexr_tuning.sprint_max_batch):
  // with fewer frames than the chip has row slots a wave's four rows share one frame's term loop instead of idling
  const int64_t sprint_max = m->tune.sprint_max_batch < 0 ? 2048 : m->tune.sprint_max_batch;  // (measured: it wins up to ~2 048 frames = one wave per frame on every SIMD pair, profiles/r05_sprint_one_frame_per_wave.txt)
  // (tail_launch: the second launch of a large batch, over the device-built list of the frames the first left unfinished)
  // (frame SEQUENCES included: a wave then walks one sequence's T frames with its four rows -- the offline retargeting of one
  // recorded hand is exactly "three rows idle")
  // (... and the buckets of a small FLEET batch: kp.B is then the whole batch, an upper bound of the bucket the kernel reads from
  // device memory -- several robots following one hand, hand_robot_viewer.py:134-181, is a batch of a few rows per model)
  const bool sprint = tail_launch || (kp.B <= sprint_max && !kp.screen && kp.n_comp == 1);
  const int fpw = sprint ? 1 : 4;  // frames per wave
  {
    const bool ladder = sprint && (tail_launch || m->tune.sprint_ladder != 0);  // (-1: policy = on; the tail launch always)
    const float mu_on[4] = {0.03f, 0.3f, 3.f, 30.f};
    for (int r = 0; r < 4; ++r) kp.sprint_mu[r] = ladder ? mu_on[r] : 1.f;
    // THE LADDER TAKES NO BLIND STEPS (round 6).  "A nearly undamped Newton step below blind_tol is the answer" is safe on the
    // four-frames-per-wave iteration, where lambda <= lambda0 is only reached through a run of well-predicted steps; on the ladder
    // the least damped row sits at 0.03 x lambda from the first pass on, and the rule ended frames 1e-4 ... 1.5e-3 rad short of
    // the minimum (offline Panda frame 1849 after 2 passes, LEAP DexPilot frame 419 after 5: tools/ladder_probe.py,
    // profiles/r06_ladder_probe.txt; the B = 2 048 parity table's certification caught both).  Every step is verified by a
    // pass at the new point; the pass that confirms convergence costs kinematics + value only (dexr_wide.hpp).
    if (ladder) kp.blind_tol = 0.f;
  }
  const int64_t tiles = (kp.B + fpw - 1) / fpw;
  // waves per SIMD: 2 (256 VGPRs; 15-17 KB of LDS per wave); the 16-row joint grid is built for 3 (168 VGPRs, 11.8 KB):
  // LEAP DexPilot 1.24 -> 1.14 ms, Allegro DexPilot 0.84 -> 0.75 ms
  // (the 24-row grid at three waves per SIMD -- 168 VGPRs, 85 registers spilled -- measured 34-40 % SLOWER in round 4, with the
  // LDS slot shrunk to make room for it: DESIGN.md section 4)
  // (one frame per wave: the 32-row grid is built for one wave per SIMD -- dexr_wide_s_* in _build.py -- the others for two)
  const int occ = sprint ? ((!m->wide_mimic && m->wbucket > 24) ? 1 : 2) : ((!m->wide_mimic && m->wbucket == 16) ? 3 : 2);
  int64_t resident = (int64_t)m->n_cu * 4 * occ;
  if (m->tune.resident_waves > 0) resident = m->tune.resident_waves;
  int64_t per_comp = (resident + kp.n_comp - 1) / kp.n_comp;
  if (per_comp > tiles) per_comp = tiles;
  const int64_t waves = per_comp * kp.n_comp;
  const int64_t blocks = (waves + wpb - 1) / wpb;
  if (blocks > 0x7fffffffLL) return fail(DEXR_ERR_INVALID, "batch too large for one launch");
  kp.q0 = (uint32_t)(per_comp * fpw);
  const unsigned slot = m->qnext.fetch_add(1u) % dexr_model::QSLOTS;
  kp.queue = m->d_queue + (size_t)slot * kp.n_comp;
  // The queue counter numbers the frames from q0 on.  When the static tiles already cover the batch (q0 >= B: every small
  // launch, the one-frame-per-call regime in particular) whatever a row draws is >= q0 >= B -- "dry" -- for ANY counter value, so
  // the reset (a fill kernel + its launch: ~4 us of a 37 us one-frame call) is skipped; the counters only ever grow, and a
  // launch that does hand out frames through the queue resets its slot as before.
  if ((int64_t)kp.q0 < kp.B) {
    hipError_t qe = hipMemsetAsync(kp.queue, 0, (size_t)kp.n_comp * sizeof(unsigned), st);
    if (qe != hipSuccess) return fail(DEXR_ERR_HIP, "queue reset failed: %s", hipGetErrorString(qe));
  }
#ifdef DEXR_WIDE_PROF
  static double* wprof = nullptr;  // profiling build only: stage cycles of wave 0 (dexr_wide.hpp WPROF_*)
  if (!wprof) (void)hipMalloc((void**)&wprof, 12 * sizeof(double));
  (void)hipMemsetAsync(wprof, 0, 12 * sizeof(double), st);
  kp.g64out = wprof;
#endif
  dexr::wide_launch_fn fn = m->wide_mimic ? (sprint ? (m->wide_modchol ? dexr::launch_wide_s_mc_16 : dexr::launch_wide_s_m_16)
                                                    : (m->wide_modchol ? dexr::launch_wide_mc_16 : dexr::launch_wide_m_16))
                            : sprint      ? dexr::find_wide_sprint_launcher(m->wbucket)
                                          : dexr::find_wide_launcher(m->wbucket);
  if (!fn) return fail(DEXR_ERR_UNSUPPORTED, "no sixteen-lane kernel for bucket %d", m->wbucket);
  hipError_t e = fn(kp, m->d_wide, dim3((unsigned)blocks), dim3(64 * wpb), per_wave * wpb, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
#ifdef DEXR_WIDE_PROF
  {
    double h[12];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, wprof, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[9] = {"hand-out", "fk", "terms", "term loop", "2nd order", "accept/active set", "factor+solve", "step", "retire"};
    fprintf(stderr, "[wprof] B=%lld passes(wave 0)=%.0f cycles per pass:", (long long)kp.B, h[11]);
    for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.0f |", names[i], h[11] > 0 ? h[i] / h[11] : 0.0);
    fprintf(stderr, "\n");
  }
#endif
  return DEXR_OK;
}

// LONGEST-FIRST ORDERING.  A launch over many more frames than the GPU holds at once (2 waves x 4 frames per SIMD) is
// bound by slow frames the queue hands out late: Shadow DexPilot, 65 536 frames = 1.08 ms of throughput + 0.83 ms of
// tail.  For DexPilot models the objective at the start point predicts the slow frames (top 5 %
// by F(x0): 91 % of the frames with >= 15 iterations), so: (1) a screening launch of the same kernel evaluates F(x0)
// (kinematics + terms, ~15 % of one solver pass per frame) and sums it, (2) frames above 1.3 x the batch mean get key 0,
// the rest key 1, (3) the fleet bucketing kernels turn the keys into an index list, (4) the solve launch walks that
// list.  Everything is stream-ordered on the caller's stream; the workspace is one of LSLOTS per-model buffers.
// MEASURED (65 536 frames): with the frames sorted by their true iteration counts (a perfect predictor, tools/
// prof_config.py DEXR_TOOL_LPT) Shadow DexPilot 1.87 -> 1.35-1.48 ms and LEAP DexPilot 1.21 -> 0.84-0.91 ms; with this
// predictor the screening + ordering launches cost 0.15-0.25 ms and the frames it misses still run 20-34 passes:
// Shadow 1.91 -> 2.04 ms, LEAP 1.23 -> 1.53 ms at ratio 2.  Hence opt-in (dexr_tuning.longest_first = 1), off by default;
// position models have no usable predictor at all (F(x0), |g|, first step, curvature: <= 45 % of the slow frames in the
// top 20 %).
// Round 4, second session: for DexPilot models the PROJECTION STATE predicts the slow frames better than F(x0) and costs
// no screening launch (dexr_aux.hip: dexpilot_key_kernel -- "a projection bit changes in this frame": 3 % of the frames, 94 %
// of those with >= 24 passes; "a projection is active": 14 % / 99.8 %): one elementwise kernel + the bucketing kernels
// (~25 us) put those frames at the front of the index list.  That IS the default for DexPilot batches large enough for the
// tail to matter (longest_first = -1: automatic; 0: never; 1: the F(x0) screening above; 2: the state keys at any size).
size_t lpt_slot_bytes(int64_t B) {
  const size_t b = (size_t)B;
  return 256 + b * sizeof(float) + b * sizeof(int32_t) + (dexr_fleet_ws_ints() + b) * sizeof(int32_t);
}
int lpt_slot_grow(dexr_model::LptSlot& sl, size_t bytes) {
  if (sl.bytes >= bytes) return DEXR_OK;
  if (sl.buf) {
    if (sl.done) HIP_TRY(hipEventSynchronize(sl.done));
    HIP_TRY(hipFree(sl.buf));
    sl.buf = nullptr;
    sl.bytes = 0;
  }
  HIP_TRY(hipMalloc(&sl.buf, bytes));
  sl.bytes = bytes;
  return DEXR_OK;
}

// TAIL LAUNCH (round 5; dexr_tuning.tail_passes).  A launch over many more frames than the chip holds is bound by its slowest
// frames: 2-6 % of the frames of a tracking batch need two to six times the mean number of passes, and they finish on a chip
// that is otherwise idle.  The first launch therefore stops every frame after P passes; the frames left with status MAXITER are
// listed on the device (one elementwise kernel + the fleet's bucketing kernels) and a second launch in the one-frame-per-wave
// shape (dexr_wide.hpp SPRINT) continues each from its accepted point with the ladder of damping values -- a pass costs less
// there and the ladder needs about half as many.  Everything is stream-ordered on the caller's stream; the workspace is one of
// the model's LSLOTS buffers (dexr_model_reserve sizes them).
int launch_wide_tail(const dexr_model* m, dexr::KernelParams kp, int P, hipStream_t st) {
  const size_t B = (size_t)kp.B;
  const size_t bytes = lpt_slot_bytes(kp.B);
  dexr_model::LptSlot& sl = m->lpt[m->lnext.fetch_add(1u) % dexr_model::LSLOTS];
  if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  else HIP_TRY(hipStreamWaitEvent(st, sl.done, 0));
  { const int rc = lpt_slot_grow(sl, bytes); if (rc != DEXR_OK) return rc; }
  unsigned char* base = static_cast<unsigned char*>(sl.buf);
  int32_t* own_status = reinterpret_cast<int32_t*>(base + 256);  // (the screening values' place: unused on this path)
  int32_t* key = reinterpret_cast<int32_t*>(base + 256 + B * sizeof(float));
  int32_t* ws = key + B;
  dexr::KernelParams k1 = kp;
  k1.max_iter = P;
  if (!k1.status) {
    k1.status = own_status;
    HIP_TRY(hipMemsetAsync(own_status, 0, B * sizeof(int32_t), st));
  }
  int rc = launch_wide_once(m, k1, st);
  if (rc != DEXR_OK) return rc;
  hipError_t e = dexr_tail_list_launch(kp.B, k1.status, kp.fval, key, ws, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "tail list kernels failed: %s", hipGetErrorString(e));
  dexr::KernelParams k2 = kp;
  k2.status = k1.status;
  k2.perm = ws + dexr_fleet_ws_ints();
  k2.bucket = ws + 2 * DEXR_FLEET_MAX_MODELS;
  k2.x0 = kp.qout;  // continue from the accepted point the first launch wrote (the regularisation target stays `last`)
  k2.iters_base = P;
  rc = launch_wide_once(m, k2, st, true);
  HIP_TRY(hipEventRecord(sl.done, st));
  return rc;
}

int launch_wide(const dexr_model* m, dexr::KernelParams kp, hipStream_t st) {
  const int want = m->tune.longest_first;
  const bool plain = !kp.perm && !kp.bucket && kp.T == 0 && kp.n_comp == 1;
  {
    // (policy: OFF.  Measured, 65 536 tracking frames, profiles/r05_tail_launch.txt: LEAP position 1.106 ms in one launch,
    // 1.18-1.29 ms with caps of 16 ... 6 passes; Shadow DexPilot 1.158 (hard frames first) -> 1.24-1.33.  The capped main launch is
    // throughput-bound and gets barely shorter, while the handed-over frames then run at a quarter of the occupancy, after it.)
    const int P = m->tune.tail_passes < 0 ? 0 : m->tune.tail_passes;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (plain && P > 0 && kp.max_iter > P && !kp.x0 && !kp.screen && kp.B >= 16384 &&
        hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone)
      return launch_wide_tail(m, kp, P, st);
  }
  // (automatic: from 9 variables on -- measured on all 13 DexPilot configs, 65 536 frames: Shadow 1.35 -> 1.11 / 1.28 -> 1.07 ms,
  // LEAP 0.90 -> 0.78 / 0.87 -> 0.78, Allegro 0.71 -> 0.60 / 0.63 -> 0.62, SVH 0.93 -> 0.77 / 1.13 -> 1.07; the six-variable
  // Ability / Inspire hands lose 2-7 % to the ~25 us of ordering, their launches are not tail-bound)
  const bool by_state = plain && kp.kind == DEXR_KIND_DEXPILOT &&
                        (want == 2 || (want < 0 && kp.n_opt >= 9 && kp.B >= 4 * (int64_t)m->n_cu * 4 * 2 * 4));
  const bool on = plain && (want == 1 || by_state);
  if (!on) return launch_wide_once(m, kp, st);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return launch_wide_once(m, kp, st);  // graph capture: no allocation / cross-stream fencing inside a captured region --
                                         // a captured graph runs the frames in natural order (same answers, the eager
                                         // call's schedule is the faster one for tail-bound batches; documented in dexr.h)
  const size_t B = (size_t)kp.B;
  const size_t bytes = lpt_slot_bytes(kp.B);
  dexr_model::LptSlot& sl = m->lpt[m->lnext.fetch_add(1u) % dexr_model::LSLOTS];
  if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  else HIP_TRY(hipStreamWaitEvent(st, sl.done, 0));  // the slot's previous user (possibly another stream) has finished
  // A slot that dexr_model_reserve(B) sized is used as it is: no allocation, no host synchronisation on this path.  Without a
  // reservation the slot grows here -- hipMalloc on the first ordered call, hipEventSynchronize + hipFree + hipMalloc when a
  // larger batch arrives (the one place a `_dev` entry point may block; call dexr_model_reserve to rule it out).
  { const int rc = lpt_slot_grow(sl, bytes); if (rc != DEXR_OK) return rc; }
  unsigned char* base = static_cast<unsigned char*>(sl.buf);
  float* sum = reinterpret_cast<float*>(base);
  float* f0 = reinterpret_cast<float*>(base + 256);
  int32_t* key = reinterpret_cast<int32_t*>(base + 256 + B * sizeof(float));
  int32_t* ws = key + B;
  int rc = DEXR_OK;
  if (by_state && want != 1) {
    hipError_t e = dexr_dexpilot_order_launch(kp.B, kp.kpts, kp.ref, kp.state, kp.n_kp, kp.n_ref, kp.h_task, kp.h_origin, kp.num_fingers,
                                              kp.project_dist, kp.escape_dist, key, ws, st);
    if (e != hipSuccess) return fail(DEXR_ERR_HIP, "ordering kernels failed: %s", hipGetErrorString(e));
  } else {
    HIP_TRY(hipMemsetAsync(sum, 0, 256, st));
    dexr::KernelParams ks = kp;
    ks.screen = f0;
    ks.screen_sum = sum;
    rc = launch_wide_once(m, ks, st);
    if (rc != DEXR_OK) return rc;
    hipError_t e = dexr_lpt_order_launch(kp.B, f0, sum, 1.3f, key, ws, st);
    if (e != hipSuccess) return fail(DEXR_ERR_HIP, "ordering kernels failed: %s", hipGetErrorString(e));
  }
  kp.perm = ws + dexr_fleet_ws_ints();
  rc = launch_wide_once(m, kp, st);
  HIP_TRY(hipEventRecord(sl.done, st));
  return rc;
}

// Root-to-leaf chains of every component's kinematic tree (one per lane of a 16-lane row) and the revolute ancestors
// of every joint, from the depth-first restore / save encoding of the tables (dexr_tables.h).
bool build_wide_tables(dexr_model* m) {
  m->wide_tabs.clear();
  m->wide_mimic = m->has_mimic;
  m->wbucket = m->bucket < 16 ? 16 : m->bucket;
  if (m->bucket < 8 || m->h.kind == DEXR_KIND_FKONLY) return false;
  if (m->has_mimic && (m->max_vars > 16 || m->max_joints > 32)) return false;
  for (const dexr_comp_table& c : m->comps) {
    dexr::WideTable w;
    std::memset(&w, 0, sizeof(w));
    std::memset(w.chain, 0xFF, sizeof(w.chain));
    const int nj = c.n_joint;
    if (c.n_term > 16 || c.n_frame > 16) return false;
    int parent[DEXR_MAXJ], slot_owner[DEXR_NSLOT + 1];
    bool has_child[DEXR_MAXJ];
    for (int s = 0; s <= DEXR_NSLOT; ++s) slot_owner[s] = -1;
    for (int k = 0; k < nj; ++k) {
      has_child[k] = false;
      if (c.src_kind[k] != DEXR_SRC_OPT && c.src_kind[k] != DEXR_SRC_FIXED && c.src_kind[k] != DEXR_SRC_MIMIC) return false;
      const int rs = c.restore[k];
      if (rs == -2) parent[k] = -1;
      else if (rs >= 0) { if (rs > DEXR_NSLOT) return false; parent[k] = slot_owner[rs]; }
      else parent[k] = k - 1;
      if (c.save[k] >= 0) { if (c.save[k] > DEXR_NSLOT) return false; slot_owner[c.save[k]] = k; }
    }
    for (int k = 0; k < nj; ++k)
      if (parent[k] >= 0) has_child[parent[k]] = true;
    bool published[DEXR_MAXJ];
    for (int k = 0; k < nj; ++k) {
      published[k] = false;
      uint32_t anc = 0;
      for (int j = k; j >= 0; j = parent[j])
        if (c.jtype[j] == DEXR_JOINT_REVOLUTE) anc |= 1u << j;
      w.anc_rev[k] = anc;
    }
    int n_chain = 0, depth = 0;
    for (int k = 0; k < nj; ++k) {
      if (has_child[k]) continue;
      if (n_chain == 16) return false;
      int path[DEXR_MAXJ], len = 0;
      for (int j = k; j >= 0; j = parent[j]) path[len++] = j;
      if (len > 16) return false;
      for (int s = 0; s < len; ++s) {
        const int j = path[len - 1 - s];
        w.chain[n_chain][s] = (uint8_t)(j | (published[j] ? 0 : 0x80));
        published[j] = true;
      }
      if (len > depth) depth = len;
      ++n_chain;
    }
    w.n_chain = n_chain;
    w.depth = depth;
    std::memset(w.fam, 0xFF, sizeof(w.fam));
    if (m->has_mimic) {
      // joint families of the variables, the variable's own joint first
      int fam_n[16] = {0};
      for (int v = 0; v < c.n_var; ++v) w.fam[v][fam_n[v]++] = (uint8_t)c.var_joint[v];
      for (int k = 0; k < nj; ++k) {
        const int v = c.var[k];
        if (v < 0 || c.var_joint[v] == k) continue;
        if (fam_n[v] == 3) return false;
        w.fam[v][fam_n[v]++] = (uint8_t)k;
      }
      for (int v = 0; v < c.n_var; ++v) w.fam_max = fam_n[v] > w.fam_max ? fam_n[v] : w.fam_max;
      // second-order pairs (k, revolute ancestor-or-self j), bucketed by the lane (vhi mod 4, vlo mod 4) that owns
      // the target entry (vhi, vlo) of the variable grid
      struct PairRec { int lane; uint32_t word; };
      std::vector<PairRec> recs;
      for (int k = 0; k < nj; ++k) {
        if (c.var[k] < 0) continue;
        for (int j = k; j >= 0; j = parent[j]) {
          if (c.jtype[j] != DEXR_JOINT_REVOLUTE || c.var[j] < 0) continue;
          const int vk = c.var[k], vj = c.var[j];
          const int vhi = vk > vj ? vk : vj, vlo = vk > vj ? vj : vk;
          const int i = vhi >> 2, jj = vlo >> 2;
          const uint32_t dbl = (j != k && vk == vj) ? 1u : 0u;
          recs.push_back({(vhi & 3) * 4 + (vlo & 3), (uint32_t)k | ((uint32_t)j << 5) | ((uint32_t)(i * (i + 1) / 2 + jj) << 10) | (dbl << 14)});
        }
      }
      if (recs.size() > 128) return false;
      int n = 0;
      for (int ln = 0; ln < 16; ++ln) {
        w.pair_off[ln] = (uint8_t)n;
        for (const PairRec& r : recs)
          if (r.lane == ln) w.pair[n++] = r.word;
        const int cnt = n - w.pair_off[ln];
        if (cnt > w.pair_max) w.pair_max = cnt;
      }
      w.pair_off[16] = (uint8_t)n;
    }
    m->wide_tabs.push_back(w);
  }
  return true;
}

#ifndef DEXR_GEN_LAM_JUMP
#define DEXR_GEN_LAM_JUMP 0.03f  // (measured, 16 384 tracking frames: arm + hand 10.3 -> 7.2 passes, -11 % time; Shadow DexPilot on generic tables 9.8 -> 8.2, -15 %; LEAP position -3 %; larger jumps leave frames over-damped up to max_iter -- profiles/r06_general_kernel_damping_jump.txt)
#endif
int launch_gen_model(const dexr_model* m, int mode, dexr::KernelParams kp, hipStream_t st) {
  const size_t lds = dexr::gen_lds_bytes(m->gen_tab);
  if (lds > 160 * 1024) return fail(DEXR_ERR_UNSUPPORTED, "model needs %zu B of LDS per frame", lds);
  // one wave per frame; as many resident waves as the LDS allows (at most 8 per CU), persistent over the batch
  int64_t per_cu = (int64_t)((160 * 1024) / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  int64_t blocks = (int64_t)m->n_cu * per_cu;
  if (blocks > kp.B) blocks = kp.B;
#ifdef DEXR_GEN_PROF
  static double* gprof = nullptr;  // profiling build only: stage cycles of block 0 (dexr_gen.hpp GPROF_*)
  if (mode == dexr::MODE_SOLVE) {
    if (!gprof) (void)hipMalloc((void**)&gprof, 12 * sizeof(double));
    (void)hipMemsetAsync(gprof, 0, 12 * sizeof(double), st);
    kp.g64out = gprof;
  }
#endif
  kp.lam_jump = m->lam_jump_user >= 0.f ? m->lam_jump_user : DEXR_GEN_LAM_JUMP;  // (x mean diag of the free block, dexr_gen.hpp)
  hipError_t e = dexr::launch_gen(mode, kp, m->gen_tab, dim3((unsigned)blocks), lds, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
#ifdef DEXR_GEN_PROF
  if (mode == dexr::MODE_SOLVE) {
    double h[12];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, gprof, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"kinematics", "terms", "term loop", "2nd order", "active set + matrix", "factorisation", "solves", "step + pred"};
    fprintf(stderr, "[gprof] B=%lld blocks=%lld lds=%zu passes(block 0)=%.0f cycles per pass:", (long long)kp.B, (long long)blocks, lds, h[11]);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f |", names[i], h[11] > 0 ? h[i] / h[11] : 0.0);
    fprintf(stderr, "\n");
  }
#endif
  return DEXR_OK;
}

#ifndef DEXR_RED_SMALL_BATCH
#define DEXR_RED_SMALL_BATCH 16384
#endif
int launch(const dexr_model* m, int mode, int f64, dexr::KernelParams kp, hipStream_t st) {
  if (kp.B <= 0) return DEXR_OK;
  if (m->gen) return launch_gen_model(m, mode, kp, st);
  // fleet buckets / frame sequences / padded rows need the kernels with extended addressing (KernelParams)
  const bool ext = kp.perm != nullptr || kp.bucket != nullptr || kp.T > 0 || kp.ld != kp.n_opt;
  if (mode == dexr::MODE_SOLVE && !f64 && selected_family(m) != FAM_REGISTER) {
    Family fam = selected_family(m);
    // THE REDUCED-VARIABLE KERNEL IS A THROUGHPUT KERNEL (round 6).  It holds a frame per LANE and walks rolled joint loops: a pass
    // of a lone wave takes ~25 us, whatever the batch.  The mimic vector models it serves by policy (Schunk SVH) also fit the
    // sixteen-lane kernel's variable grid, which is 2 x faster up to 16 384 frames and 1.5 x slower at 65 536 (one MI355X, tracking
    // frames: B = 1: 108 -> 51 us, 2 048: 309 -> 194, 16 384: 560 -> 327, 65 536: 708 vs 1 067; profiles/r06_svh_vector_reduced_vs_wide.txt)
    // -- and the reference's own benchmark (profile_online_retargeting.py, one frame per call) had this row at 3 266 fps on the GPU
    // against 5 968 for the CPU port.  So batches of up to DEXR_RED_SMALL_BATCH frames of such a model take the sixteen-lane
    // kernel (automatic policy only: dexr_tuning.kernel = DEXR_KERNEL_REDUCED keeps the reduced kernel at every size).
    // Like sprint_max_batch this makes the ITERATION a function of the batch size: same minimiser, same tolerance, and the
    // 16 384 / 16 385 straddle test (tests/test_gpu_all_configs.py) pins how often a multi-modal frame settles elsewhere.
    if (fam == FAM_RED && m->wide_ok && m->tune.kernel == DEXR_KERNEL_AUTO && kp.B <= DEXR_RED_SMALL_BATCH && !kp.screen) fam = FAM_WIDE;
    kp.lam_jump = family_lam_jump(m, fam);
    kp.lam_fastdec = family_lam_fastdec(m, fam);
    return fam == FAM_RED ? launch_red(m, kp, st) : fam == FAM_WIDE ? launch_wide(m, kp, st)
         : fam == FAM_QUAD ? launch_quad(m, kp, st) : launch_big(m, kp, st);
  }
  kp.lam_jump = family_lam_jump(m, FAM_REGISTER);
  if (m->bucket == 32 && mode == dexr::MODE_SOLVE) f64 = 1;  // see find_launcher: bucket 32 is float64 only
  const size_t real_sz = f64 ? 8 : 4;
  // (the tip pass keeps the placements of joints 1..3 in 64 more floats of the wave's LDS, dexr_tip.hpp)
  const bool tip_kernel = m->tip && m->chain && m->bucket == 4 && mode == dexr::MODE_SOLVE;
  // (... and the float64 tip kernel parks the accepted model -- 10 + 4 + 4 doubles per lane -- in LDS: dexr_kernel.hpp PARK)
  const size_t per_wave = 64 * real_sz * (size_t)(3 * m->lds_frames + 4 * m->lds_terms + (tip_kernel ? 1 : 0) + ((tip_kernel && f64) ? 18 : 0));
#ifndef DEXR_WPB
#define DEXR_WPB 4  // waves per block of the register kernels (the kernels never synchronise across waves)
#endif
  int wpb = DEXR_WPB;
#ifndef DEXR_TIP_WPB_BIG
#define DEXR_TIP_WPB_BIG 8
#endif
  // float32 tip kernel, launches that fill the chip four waves deep (>= 4 096 waves): blocks of 8 waves.  The workgroup
  // dispatcher places the chip's first two waves per SIMD within ~4 us and needs another ~12 us for the third and fourth
  // (tools/wave_trace.sh); with half as many workgroups to place the launch of 65 536 Allegro frames takes 45.0 instead of
  // 47.5 us (16 waves per block: 47.9; 1 wave per block: 52.3).  Smaller launches are faster in blocks of 4 (16 384 frames:
  // 27.2 vs 33.3 us), so the policy is by size.
  if (tip_kernel && !f64 && (kp.B + 63) / 64 * kp.n_comp >= 4096) wpb = DEXR_TIP_WPB_BIG;
  // (float64 tip kernel: 14.8 KB per wave with the parked model -- blocks of FOUR waves all the same (59 KB, two blocks per CU): the
  // four finger components of a tile of frames read the same keypoint / last_qpos lines, and in blocks of two they sat on different
  // CUs and XCDs -- FETCH_SIZE 38.9 MB per launch against 24.9 MB algorithmic, the float32 kernel's blocks of eight fetch 25.1)
  const size_t block_lds_cap = (tip_kernel && f64) ? 64 * 1024 : 48 * 1024;
  while (wpb > 1 && per_wave * wpb > block_lds_cap) wpb >>= 1;
  if (per_wave > 64 * 1024) return fail(DEXR_ERR_UNSUPPORTED, "component needs %zu B of LDS per wave", per_wave);
  const int64_t tiles = (kp.B + 63) / 64;
  int64_t waves = tiles * kp.n_comp;
  if (mode == dexr::MODE_SOLVE && m->bucket <= 8) {
    // persistent lanes (small components): a resident set of waves pulls frames from per-component queues
    int occ = m->chain ? 4 : (m->bucket <= 4 ? 3 : 2);  // waves per SIMD the kernels' register budgets allow
    if (f64) occ = tip_kernel ? 2 : 1;                  // (float64: the tip pass is built for two, the generic kernels hold one)
    if (m->tune.persist_occ > 0) occ = m->tune.persist_occ;
    const int64_t resident = (int64_t)m->n_cu * 4 * occ;
    const int64_t per_comp = (resident + kp.n_comp - 1) / kp.n_comp;
    // Few frames per lane: one 64-frame tile per wave (no queue).  Many frames per lane: a resident set of waves
    // drains the queue in chunks, which evens out the different iteration counts of individual frames.  Measured
    // (profiles/r01_term_sweep.txt, Allegro vector): tile mode is faster up to 262 144 frames (0.20 vs 0.25 ms), the
    // queue wins at 1 M (0.61 vs 0.67 ms).
    const int64_t persist_from = m->tune.persist_from;
    kp.qchunk = 0;  // tile mode: no queue traffic at all
    if (tiles >= persist_from * per_comp) {
      waves = per_comp * kp.n_comp;
      kp.q0 = (uint32_t)(per_comp * 64);
      kp.qchunk = m->tune.qchunk > 0 ? (uint32_t)m->tune.qchunk : 256u;
    }
    if (kp.qchunk) {
      const unsigned slot = m->qnext.fetch_add(1u) % dexr_model::QSLOTS;
      kp.queue = m->d_queue + (size_t)slot * kp.n_comp;
      hipError_t qe = hipMemsetAsync(kp.queue, 0, (size_t)kp.n_comp * sizeof(unsigned), st);
      if (qe != hipSuccess) return fail(DEXR_ERR_HIP, "queue reset failed: %s", hipGetErrorString(qe));
    }
  }
  const int64_t blocks = (waves + wpb - 1) / wpb;
  if (blocks > 0x7fffffffLL) return fail(DEXR_ERR_INVALID, "batch too large for one launch");
  dexr::launch_fn fn = dexr::find_launcher(m->bucket, f64, mode, m->chain, ext, m->tip);
  if (!fn) return fail(DEXR_ERR_UNSUPPORTED, "no kernel for bucket %d / f64=%d / mode=%d", m->bucket, f64, mode);
#ifdef DEXR_SMALL_PROF
  static double* sprof = nullptr;  // profiling build only: stage cycles of wave 0 (dexr_kernel.hpp SPROF_*)
  const bool sprof_on = mode == dexr::MODE_SOLVE && m->bucket <= 8;
  if (sprof_on) {
    if (!sprof) (void)hipMalloc((void**)&sprof, 12 * sizeof(double));
    (void)hipMemsetAsync(sprof, 0, 12 * sizeof(double), st);
    kp.g64out = sprof;
  }
#endif
#ifdef DEXR_WAVE_TRACE
  // profiling build only (tools/wave_trace.sh): 24 doubles per wave, dumped to /tmp/dexr_wave_trace.bin after every launch
  static double* wtrace = nullptr;
  static size_t wtrace_cap = 0;
  const bool wtrace_on = mode == dexr::MODE_SOLVE && m->bucket <= 8;
  const size_t wtrace_n = (size_t)blocks * wpb * 24;
  if (wtrace_on) {
    if (wtrace_n > wtrace_cap) {
      if (wtrace) (void)hipFree(wtrace);
      (void)hipMalloc((void**)&wtrace, wtrace_n * sizeof(double));
      wtrace_cap = wtrace_n;
    }
    (void)hipMemsetAsync(wtrace, 0, wtrace_n * sizeof(double), st);
    kp.g64out = wtrace;
  }
#endif
  hipError_t e = fn(kp, dim3((unsigned)blocks), dim3(64 * wpb), per_wave * wpb, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
#ifdef DEXR_WAVE_TRACE
  if (wtrace_on) {
    std::vector<double> h(wtrace_n);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), wtrace, wtrace_n * sizeof(double), hipMemcpyDeviceToHost);
    if (FILE* f = fopen("/tmp/dexr_wave_trace.bin", "wb")) {
      fwrite(h.data(), sizeof(double), wtrace_n, f);
      fclose(f);
    }
  }
#endif
#ifdef DEXR_SMALL_PROF
  if (sprof_on) {
    double h[12];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, sprof, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"hand-out", "step", "fk", "residuals", "accept", "retire"};
    fprintf(stderr, "[sprof] B=%lld passes(wave 0)=%.0f cycles per pass:", (long long)kp.B, h[11]);
    for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.0f |", names[i], h[11] > 0 ? h[i] / h[11] : 0.0);
    fprintf(stderr, "\n");
  }
#endif
  return DEXR_OK;
}

// Default launch / damping parameters of a model (measured, tools/term_sweep.py): a rejected step raises lambda at
// least to lam_jump x the curvature scale instead of creeping up by x2, x4, ...; small components also drop it by 10x
// (not 3x) after a step the model predicted well.  Allegro vector, 65 536 frames: 0.143 -> 0.119 ms; Shadow DexPilot:
// 19.4 -> 15.4 ms.
void default_tuning(dexr_model* m) {
  dexr_tuning& t = m->tune;
  std::memset(&t, 0, sizeof(t));
  t.struct_size = (uint32_t)sizeof(dexr_tuning);
  t.kernel = DEXR_KERNEL_AUTO;
  t.chain = 1;
  t.persist_occ = 0;
  t.persist_from = 8;
  t.qchunk = 256;
  t.resident_waves = 0;
  t.max_blind = 8;
  t.stall_from = 2;
  t.stall_ratio = 0.9f;
  t.stall_cap = 20.f;
  t.lam_jump = m->bucket <= 8 ? 3.0f : 0.3f;
  t.lam_fastdec = m->bucket <= 8 ? 0.1f : 0.f;  // (the sixteen-lane kernel has its own default: family_lam_fastdec)
  t.lam_recover = 0.f;
  t.floor_scale = 1e-12f;
  t.step_cap = 0.3f;
  // small components: a verified, undamped Newton step below 100 tol (2e-4 rad) leaves an error of ~C s^2 < 1e-6 rad
  // (tools/lm_lab.py: max 8.6e-7 over 1 863 frames) and saves the confirming pass: mean 4.3 -> 3.9 passes per frame
  // (larger components: 10 tol.  Measured with tools/all_configs.py + DEXR_TOOL_KNOBS: 30 tol is 2-3 % faster per launch, but a
  // far-start LEAP DexPilot frame then stops 2e-3 rad short of its stationary point -- tests/test_gpu_all_configs.py)
  t.blind_tol_scale = m->bucket <= 8 ? 100.f : 10.f;
  t.pivot_rule = -1;
  t.longest_first = -1;
  t.fork_streams = -1;
  t.sprint_max_batch = -1;
  t.sprint_ladder = -1;
  t.tail_passes = -1;
}

// A model in the generic table format (dexr_tables.h): validate every index the general kernel will follow, upload the
// arrays in one block and point the kernel's GenTab at them.
int create_generic(const dexr_model_header& h, const char* body, size_t nbytes, dexr_model** out) {
  if (h.version != DEXR_TABLE_VERSION) return fail(DEXR_ERR_INVALID, "table version %u, library expects %u", h.version, DEXR_TABLE_VERSION);
  if (nbytes < sizeof(dexr_gen_header)) return fail(DEXR_ERR_INVALID, "blob shorter than the generic-table header");
  dexr_gen_header gh;
  std::memcpy(&gh, body, sizeof(gh));
  if (gh.magic != DEXR_GEN_MAGIC) return fail(DEXR_ERR_INVALID, "n_comp = 0 but no generic table follows the header");
  const int nj = gh.n_joint, nf = gh.n_frame, nt = gh.n_term, nv = gh.n_var, nfam = gh.n_fam;
  if (nj < 0 || nj > DEXR_GEN_MAXJ || nf < 1 || nf > DEXR_GEN_MAXJ || nt < 1 || nt > DEXR_GEN_MAXJ || nv < 0 || nv > nj ||
      nfam < nv || nfam > nj || gh.max_depth < 0 || gh.max_depth > nj)
    return fail(DEXR_ERR_INVALID, "generic table sizes out of range (%d joints, %d frames, %d terms, %d variables)", nj, nf, nt, nv);
  if (h.kind != DEXR_KIND_FKONLY && nv != h.n_opt) return fail(DEXR_ERR_INVALID, "generic table has %d variables, header n_opt = %d", nv, h.n_opt);
  if (nt != h.n_ref) return fail(DEXR_ERR_INVALID, "generic table has %d terms, header n_ref = %d", nt, h.n_ref);
  if (h.kind == DEXR_KIND_DEXPILOT) {
    const int F = h.num_fingers;
    if (F < 2 || F > 8) return fail(DEXR_ERR_INVALID, "DexPilot needs 2..8 fingers (32 projection bits), got %d", F);
    if (h.n_ref != F * (F - 1) / 2 + F) return fail(DEXR_ERR_INVALID, "DexPilot n_ref=%d does not match %d fingers", h.n_ref, F);
  }
  auto pad2 = [](size_t n) { return (n + 1) & ~(size_t)1; };
  const size_t n_f64 = (size_t)nj * 12 + (size_t)nj * 3 + 2 * (size_t)nj + 2 * (size_t)nv + (size_t)nf * 3;
  const size_t n_u64 = (size_t)nf + nj;
  const size_t i32_sizes[14] = {(size_t)nj, (size_t)nj, (size_t)nj, (size_t)nj, (size_t)nj, (size_t)nv, (size_t)nv + 1,
                                (size_t)nfam, (size_t)nf, (size_t)nt, (size_t)nt, (size_t)nt, (size_t)nt, (size_t)nt};
  size_t n_i32 = 0;
  for (size_t v : i32_sizes) n_i32 += pad2(v);
  const size_t want = sizeof(gh) + 8 * (n_f64 + n_u64) + 4 * n_i32;
  if (nbytes != want) return fail(DEXR_ERR_INVALID, "generic table is %zu B, its header implies %zu B", nbytes, want);
  const char* p = body + sizeof(gh);
  const double* f64 = reinterpret_cast<const double*>(p);
  const unsigned long long* u64 = reinterpret_cast<const unsigned long long*>(p + 8 * n_f64);
  const int32_t* i32 = reinterpret_cast<const int32_t*>(p + 8 * (n_f64 + n_u64));
  const int32_t* arr[14];
  {
    size_t o = 0;
    for (int i = 0; i < 14; ++i) {
      arr[i] = i32 + o;
      o += pad2(i32_sizes[i]);
    }
  }
  const int32_t *jtype = arr[0], *parent = arr[1], *depth = arr[2], *src_idx = arr[3], *var = arr[4], *var_api = arr[5],
                *fam_off = arr[6], *fam = arr[7], *frame_joint = arr[8], *term_task = arr[9], *term_origin = arr[10],
                *term_ref = arr[11], *row_ho = arr[12], *row_ht = arr[13];
  for (int k = 0; k < nj; ++k) {
    const bool bad = (jtype[k] != DEXR_JOINT_REVOLUTE && jtype[k] != DEXR_JOINT_PRISMATIC) || parent[k] < -1 || parent[k] >= k ||
                     depth[k] != (parent[k] < 0 ? 0 : depth[parent[k]] + 1) || depth[k] > gh.max_depth || var[k] < -1 || var[k] >= nv ||
                     (var[k] < 0 && h.kind != DEXR_KIND_FKONLY && (src_idx[k] < 0 || src_idx[k] >= h.n_fixed)) ||
                     (h.kind == DEXR_KIND_FKONLY && (src_idx[k] < 0 || src_idx[k] >= h.n_q));
    if (bad) return fail(DEXR_ERR_INVALID, "generic table: joint record %d malformed", k);
  }
  for (int v = 0; v < nv; ++v) {
    if (var_api[v] < 0 || var_api[v] >= h.n_opt || fam_off[v] < 0 || fam_off[v] > fam_off[v + 1] || fam_off[v + 1] > nfam)
      return fail(DEXR_ERR_INVALID, "generic table: variable record %d malformed", v);
    for (int e = fam_off[v]; e < fam_off[v + 1]; ++e)
      if (fam[e] < 0 || fam[e] >= nj || var[fam[e]] != v) return fail(DEXR_ERR_INVALID, "generic table: family of variable %d malformed", v);
  }
  for (int f = 0; f < nf; ++f)
    if (frame_joint[f] < -1 || frame_joint[f] >= nj) return fail(DEXR_ERR_INVALID, "generic table: frame record %d malformed", f);
  std::vector<bool> seen((size_t)nt, false);
  for (int t = 0; t < nt; ++t) {
    const bool bad = term_task[t] < 0 || term_task[t] >= nf || term_origin[t] < -1 || term_origin[t] >= nf || term_ref[t] < 0 ||
                     term_ref[t] >= nt || seen[(size_t)term_ref[t]] ||
                     (gh.has_keypoint_map && (row_ht[t] < 0 || row_ht[t] >= h.n_keypoints || row_ho[t] < -1 || row_ho[t] >= h.n_keypoints));
    if (bad) return fail(DEXR_ERR_INVALID, "generic table: term record %d malformed", t);
    seen[(size_t)term_ref[t]] = true;
  }
  dexr_model* m = new (std::nothrow) dexr_model();
  if (!m) return fail(DEXR_ERR_INVALID, "out of host memory");
  m->h = h;
  m->gen = true;
  m->bucket = 0;
  m->max_joints = nj;
  m->max_vars = nv;
  default_tuning(m);
  m->has_mimic = nfam > nv;
  const size_t payload = nbytes - sizeof(gh);
  hipError_t e = hipMalloc(&m->d_gen, payload ? payload : 8);
  if (e == hipSuccess) e = hipMemcpy(m->d_gen, p, payload, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      m->n_cu = cus;
  }
  if (e != hipSuccess) {
    if (m->d_gen) (void)hipFree(m->d_gen);
    delete m;
    return fail(DEXR_ERR_HIP, "uploading tables failed: %s", hipGetErrorString(e));
  }
  const char* d = static_cast<const char*>(m->d_gen);
  const double* df = reinterpret_cast<const double*>(d);
  dexr::GenTab& g = m->gen_tab;
  g.nj = nj; g.nf = nf; g.nt = nt; g.nv = nv; g.nfam = nfam; g.max_depth = gh.max_depth; g.has_kp = gh.has_keypoint_map;
  g.X = df;
  g.axis = g.X + (size_t)nj * 12;
  g.jmul = g.axis + (size_t)nj * 3;
  g.joff = g.jmul + nj;
  g.lo = g.joff + nj;
  g.hi = g.lo + nv;
  g.frame_off = g.hi + nv;
  g.frame_anc = reinterpret_cast<const unsigned long long*>(d + 8 * n_f64);
  g.joint_anc = g.frame_anc + nf;
  const int32_t* di = reinterpret_cast<const int32_t*>(d + 8 * (n_f64 + n_u64));
  const int32_t** dst[14] = {&g.jtype, &g.parent, &g.depth, &g.src_idx, &g.var, &g.var_api, &g.fam_off, &g.fam, &g.frame_joint,
                             &g.term_task, &g.term_origin, &g.term_ref, &g.row_ho, &g.row_ht};
  {
    size_t o = 0;
    for (int i = 0; i < 14; ++i) {
      *dst[i] = di + o;
      o += pad2(i32_sizes[i]);
    }
  }
  (void)u64;
  (void)f64;
  *out = m;
  return DEXR_OK;
}

// Which float32 solve kernel serves the model.  Measured on MI355X (65 536 frames, tools/all_configs.py,
// tools/cmp_big.py): the LDS kernel wins where the register kernel needs its float64 polish launch on a large
// component (position models: Inspire 8.7 vs 27 ms, Shadow + free joints 57 vs 170 ms) and loses where it does not
// (Shadow vector 20 vs 3.8 ms).  With persistent quads the quad kernel wins for every DexPilot model without mimic
// joints (Shadow 6.3 ms vs 35.8 ms register + float64 polish; LEAP 3.8-4.4 vs 6.5 ms; Allegro 2.0-2.3 vs 2.1-2.6 ms) and
// for position models with free joints (LEAP 3.3 vs 9.5 ms LDS kernel); the register kernel stays ahead for Shadow
// vector (1.9-3.0 vs 3.2-3.9 ms).  tune.kernel overrides the policy where the chosen family supports the model.
// One dense component of 5-8 joints (Panda gripper + 6 free joints, position objective): the register kernel solves it
// in one lane per frame and needs its float64 polish launch; measured on MI355X (65 536 frames, tools/all_configs.py):
// 1.06 ms there vs the sixteen-lane kernel's figure recorded in DESIGN.md section 4.
bool wide_wins_small(const dexr_model* m) {
  return m->bucket == 8 && m->h.n_comp == 1 && (m->h.kind == DEXR_KIND_POSITION || m->h.kind == DEXR_KIND_DEXPILOT);
}

void select_kernels(dexr_model* m) {
  const dexr_model_header& h = m->h;
  const int want = m->tune.kernel;
  const size_t lds = (size_t)64 * (4 * (size_t)m->big_nh_rows + 8 * 3 * (size_t)m->lds_frames);
  // four lanes per frame: dense components of 9..24 joints without mimic joints, one fork level
  const bool quad_ok = (m->bucket == 16 || m->bucket == 24) && !m->has_mimic && m->max_slot < 1 && h.kind != DEXR_KIND_FKONLY;
  const bool big_ok = m->bucket >= 16 && h.kind != DEXR_KIND_FKONLY && m->max_slot < 2 && lds <= 160 * 1024;
  const bool quad_wins = h.kind == DEXR_KIND_DEXPILOT || h.kind == DEXR_KIND_POSITION;
  const bool big_wins = h.kind == DEXR_KIND_POSITION || m->bucket == 32;
  // reduced variables: the Hessian of the n_var <= 16 optimised variables in registers, the kinematics of the (up to
  // 32) joints in LDS.  Serves the models with mimic joints whose components outgrow the small kernels (Ability /
  // Inspire DexPilot and position models, every Schunk SVH model): their joint-space Hessian is 2-3 x the size.
  m->red_nv = m->max_vars <= 8 ? 8 : (m->max_vars <= 16 ? 16 : 0);
  const size_t red_lds = (size_t)64 * (4 * 6 * (size_t)m->max_joints + 8 * 3 * (size_t)m->lds_frames);
  const bool red_ok = m->red_nv > 0 && h.kind != DEXR_KIND_FKONLY && m->max_slot < 2 && red_lds <= 160 * 1024 && m->max_joints > 0;
  const bool red_wins = m->has_mimic && m->bucket >= 16;
  m->red = red_ok && (want == DEXR_KERNEL_REDUCED || (want == DEXR_KERNEL_AUTO && red_wins));
  // sixteen lanes per frame: measured ahead of every other family on every model it supports (Shadow DexPilot 1.9 vs
  // 6.4 ms quad, LEAP position 2.1 vs 5.0 ms quad, Shadow vector 1.2 vs 1.9-3.0 ms register, Shadow + free joints 9.8 vs
  // 23-29 ms LDS; 65 536 frames)
  // (components of 5-8 joints fit the 16-row grid too: measured policy below)
  m->wide = m->wide_ok && (want == DEXR_KERNEL_WIDE || (want == DEXR_KERNEL_AUTO && (m->bucket >= 16 || wide_wins_small(m))));
  // mimic vector models (SVH: two small components) stay on the reduced-variable kernel: 0.7-0.8 vs 0.9-1.1 ms
  if (m->wide && m->has_mimic && want == DEXR_KERNEL_AUTO && h.kind == DEXR_KIND_VECTOR && red_ok) m->wide = false;
  // (the modified-Cholesky rules are instantiated for the variable-grid kernel only)
  m->wide_modchol = m->wide && m->wide_mimic && (m->tune.pivot_rule > 0 || (m->tune.pivot_rule < 0 && h.kind == DEXR_KIND_DEXPILOT));
  m->red = m->red && !m->wide;
  m->quad = !m->wide && !m->red && quad_ok && (want == DEXR_KERNEL_QUAD || (want == DEXR_KERNEL_AUTO && quad_wins));
  m->big = !m->wide && !m->red && !m->quad && big_ok && (want == DEXR_KERNEL_LDS || (want == DEXR_KERNEL_AUTO && big_wins));
  // the quad kernel scales its damping jump by the curvature along the failed step (not by mean diag H)
  // serial-chain specialisation (LocalTab keeps LF frames / LT terms in registers): every component must be an
  // unbranched chain of exactly `bucket` revolute optimised joints with at most LF frames and LT terms
  m->chain = m->tune.chain != 0 && (h.kind == DEXR_KIND_VECTOR || h.kind == DEXR_KIND_POSITION);
  for (const dexr_comp_table& c : m->comps) {
    if (c.n_joint != m->bucket || c.n_frame > dexr::LocalTab<4>::LF || c.n_term > dexr::LocalTab<4>::LT) m->chain = false;
    for (int k = 0; k < c.n_joint && m->chain; ++k)
      if (c.restore[k] != (k == 0 ? -2 : -1) || c.save[k] != -1 || c.src_kind[k] != DEXR_SRC_OPT ||
          c.jtype[k] != DEXR_JOINT_REVOLUTE)
        m->chain = false;
  }
  if (m->bucket != 4) m->chain = false;  // only the 4-joint bucket has a chain instantiation
  // tip pass (dexr_tip.hpp): VectorOptimizer components with ONE term whose origin frame sits on the base and whose task
  // frame sits on the last joint -- every finger of the per-finger vector models (dexr_tuning.chain = 2: plain chain kernel)
  m->tip = m->chain && m->tune.chain == 1 && h.kind == DEXR_KIND_VECTOR;
  for (size_t ci = 0; ci < m->comps.size() && m->tip; ++ci) {
    const dexr_comp_table& c = m->comps[ci];
    if (c.n_term != 1) { m->tip = false; break; }
    const int ft = c.term_task[0], fo = c.term_origin[0];
    if (ft < 0 || ft >= c.n_frame || fo < 0 || fo >= c.n_frame || c.frame_joint[ft] != 3 || c.frame_joint[fo] != -1) m->tip = false;
    for (int k = 1; k < 4 && m->tip; ++k)  // the four joints are consecutive columns of last_qpos / qpos_out
      if (c.api[k] != c.api[0] + k) m->tip = false;
  }
}

void apply_options(const dexr_model* m, dexr::KernelParams& kp, const dexr_solve_options* opt) {
  dexr_solve_options o;
  dexr_default_options(&o);
  if (opt) o = *opt;
  kp.max_iter = o.max_iter > 0 ? o.max_iter : 64;
  kp.tol = o.tol > 0 ? o.tol : 2e-6f;
  kp.lam0 = o.lambda0 > 0 ? o.lambda0 : 1e-4f;
  kp.newton = o.newton;
  kp.max_blind = m->tune.max_blind;
  kp.blind_tol = m->tune.blind_tol_scale * kp.tol;
  kp.stall_from = m->tune.stall_from;
  kp.stall_ratio = m->tune.stall_ratio;
  kp.stall_cap = m->tune.stall_cap;
}

// Optional float64 polish: same kernel in double precision, started at the float32 answer (x0 = qout, in place),
// regularised towards the ORIGINAL last_qpos.  Stream-ordered after the float32 launch.
bool polish_wanted(const dexr_model* m, const dexr_solve_options* opt) {
  if (m->gen) return false;  // the general kernel is float64 throughout
  int polish = opt ? opt->polish : -1;
  if (polish < 0) polish = (m->h.kind == DEXR_KIND_POSITION || m->h.kind == DEXR_KIND_DEXPILOT) ? 24 : 0;
  if (polish == 0 || m->bucket == 32) return false;
  const int strict = opt ? opt->strict : 0;
  if ((m->red || (m->wide && m->wide_mimic)) && strict <= 0) return false;  // variable-space kernels: float64 kinematics
                                                                            // (incl. the mimic joint values) and value
  if ((m->big || m->quad || m->wide) && !(strict > 0 || (strict == 0 && m->has_mimic))) return false;
  return true;
}

// The mixed-precision kernels (float64 kinematics, float32 gradient / Hessian) need no polish -- except on models
// with mimic joints, whose objective has nearly flat valleys (net curvature ~3e-4: the indefinite second-order term
// almost cancels the regulariser) in which a float32 gradient error of 1e-7 moves the stationary point by up to
// 4e-4 rad (round 1: Ability / Inspire / SVH position models had 0.01-0.6 % of frames between 1e-4 and 4e-4 rad of
// the float64 minimiser).  Those models get the float64 polish by default (polish_wanted); strict = 1 forces it for
// every model, strict = -1 never polishes after the mixed-precision kernels.
int polish_launch(const dexr_model* m, dexr::KernelParams kp, const dexr_solve_options* opt, hipStream_t st) {
  if (!polish_wanted(m, opt)) return DEXR_OK;
  int polish = opt ? opt->polish : -1;
  if (polish < 0) polish = 24;
  kp.x0 = kp.qout;
  kp.max_iter = polish;
  kp.tol *= 0.25f;
  kp.blind_tol = 10.f * kp.tol;  // the polish pass confirms its steps down to its own (tighter) tolerance
  kp.fval = nullptr;  // fval/iters keep the float32 launch's diagnostics; status is the polish pass's verdict
  if (kp.perm) kp.status = nullptr;  // fleet batches share one status array: it keeps the first launch's verdict
  if (kp.status) HIP_TRY(hipMemsetAsync(kp.status, 0, (size_t)kp.B * sizeof(int32_t), st));
  return launch(m, dexr::MODE_SOLVE, 1, kp, st);
}

}  // namespace

// dexr_prep.hip
int dexr_prep_launch(int64_t B, const float* kp, const float* op9, float* out, float* rot, hipStream_t st);
// dexr_aux.hip
size_t dexr_fleet_ws_ints();
hipError_t dexr_fleet_bucket_launch(int n_models, int64_t B, const int32_t* model_id, int32_t* ws, hipStream_t st);
hipError_t dexr_lpt_order_launch(int64_t B, const float* f0, const float* sum, float ratio, int32_t* key, int32_t* ws, hipStream_t st);
hipError_t dexr_dexpilot_order_launch(int64_t B, const float* kpts, const float* ref, const uint32_t* state, int n_kp, int n_ref,
                                      const int32_t* h_task, const int32_t* h_origin, int F, float project_dist, float escape_dist,
                                      int32_t* key, int32_t* ws, hipStream_t st);
hipError_t dexr_seq_compose_launch(int64_t B, int T, int n_q, int n_opt, int n_fixed, const int32_t* kind,
                                   const int32_t* idx, const double* mult, const double* off, const float* qraw,
                                   const float* fixed, double alpha, int use_filter, int first_frame_initialises,
                                   double* filt, double* out, hipStream_t st);

extern "C" {

const char* dexr_last_error(void) { return g_err.c_str(); }

const char* dexr_version(void) { return "dexr 0.4 (gfx950; table v5 + generic tables)"; }

int dexr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void dexr_default_options(dexr_solve_options* opt) {
  if (!opt) return;
  opt->strict = 0;
  opt->max_iter = 64;
  opt->tol = 2e-6f;
  opt->lambda0 = 1e-4f;
  opt->newton = 1;
  opt->precision = 0;
  opt->polish = -1;
}

int dexr_model_create(const void* blob, size_t nbytes, dexr_model** out) {
  if (!blob || !out) return fail(DEXR_ERR_INVALID, "null argument");
  *out = nullptr;
  if (nbytes < sizeof(dexr_model_header)) return fail(DEXR_ERR_INVALID, "blob shorter than the header");
  dexr_model_header h;
  std::memcpy(&h, blob, sizeof(h));
  if (h.magic != DEXR_MAGIC) return fail(DEXR_ERR_INVALID, "bad magic 0x%08x", h.magic);
  if (h.version != DEXR_TABLE_VERSION) return fail(DEXR_ERR_INVALID, "table version %u, library expects %u", h.version, DEXR_TABLE_VERSION);
  if (h.kind < DEXR_KIND_VECTOR || h.kind > DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "unknown kind %d", h.kind);
  if (h.n_opt < 0 || h.n_fixed < 0 || h.n_ref < 0 || h.n_q < 0) return fail(DEXR_ERR_INVALID, "negative size in header");
  if (h.n_comp == 0 && h.comp_bytes == 0) return create_generic(h, static_cast<const char*>(blob) + sizeof(h), nbytes - sizeof(h), out);
  if (h.comp_bytes != (int32_t)sizeof(dexr_comp_table)) return fail(DEXR_ERR_INVALID, "component record is %d B, library expects %zu", h.comp_bytes, sizeof(dexr_comp_table));
  if (h.n_comp < 1 || h.n_comp > 4096) return fail(DEXR_ERR_INVALID, "n_comp=%d out of range", h.n_comp);
  if (h.kind < DEXR_KIND_VECTOR || h.kind > DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "unknown kind %d", h.kind);
  if (nbytes != sizeof(h) + (size_t)h.n_comp * sizeof(dexr_comp_table)) return fail(DEXR_ERR_INVALID, "blob size %zu does not match n_comp=%d", nbytes, h.n_comp);
  if (h.n_opt < 0 || h.n_fixed < 0 || h.n_ref < 0 || h.n_q < 0) return fail(DEXR_ERR_INVALID, "negative size in header");
  if (h.kind == DEXR_KIND_DEXPILOT) {
    if (h.num_fingers < 2 || h.num_fingers > 5) return fail(DEXR_ERR_INVALID, "DexPilot needs 2..5 fingers, got %d", h.num_fingers);
    if (h.n_comp != 1) return fail(DEXR_ERR_INVALID, "DexPilot tables must have exactly one component");
    const int F = h.num_fingers;
    if (h.n_ref != F * (F - 1) / 2 + F) return fail(DEXR_ERR_INVALID, "DexPilot n_ref=%d does not match %d fingers", h.n_ref, F);
  }
  dexr_model* m = new (std::nothrow) dexr_model();
  if (!m) return fail(DEXR_ERR_INVALID, "out of host memory");
  m->h = h;
  m->comps.resize(h.n_comp);
  std::memcpy(m->comps.data(), static_cast<const char*>(blob) + sizeof(h), (size_t)h.n_comp * sizeof(dexr_comp_table));
  int maxj = 0;
  for (const dexr_comp_table& c : m->comps) {
    if (c.n_joint < 0 || c.n_joint > DEXR_MAXJ || c.n_frame < 0 || c.n_frame > DEXR_MAXF || c.n_term < 0 ||
        c.n_term > DEXR_MAXT || c.n_base_frame < 0 || c.n_base_frame > c.n_frame) {
      delete m;
      return fail(DEXR_ERR_INVALID, "component sizes out of range");
    }
    for (int k = 0; k < c.n_joint; ++k) {
      const bool bad = c.restore[k] < -2 || c.restore[k] >= DEXR_NSLOT || c.save[k] < -1 || c.save[k] >= DEXR_NSLOT ||
                       c.fbeg[k] < 0 || c.fend[k] > c.n_frame || c.src_kind[k] < 0 || c.src_kind[k] > DEXR_SRC_DIRECT ||
                       (c.src_kind[k] == DEXR_SRC_OPT && (c.api[k] < 0 || c.api[k] >= h.n_opt)) ||
                       (c.src_kind[k] == DEXR_SRC_FIXED && (c.src_idx[k] < 0 || c.src_idx[k] >= h.n_fixed)) ||
                       (c.src_kind[k] == DEXR_SRC_MIMIC && (c.src_idx[k] < 0 || c.src_idx[k] >= c.n_joint)) ||
                       (c.src_kind[k] == DEXR_SRC_DIRECT && (c.src_idx[k] < 0 || c.src_idx[k] >= h.n_q));
      if (bad) {
        delete m;
        return fail(DEXR_ERR_INVALID, "joint record %d malformed", k);
      }
    }
    for (int t = 0; t < c.n_term; ++t) {
      if (c.term_task[t] < 0 || c.term_task[t] >= c.n_frame || c.term_origin[t] < -1 || c.term_origin[t] >= c.n_frame ||
          c.term_ref[t] < 0 || c.term_ref[t] >= h.n_ref) {
        delete m;
        return fail(DEXR_ERR_INVALID, "term record %d malformed", t);
      }
    }
    {
      bool bad = c.n_var < 0 || c.n_var > c.n_joint;
      int n_opt_joints = 0;
      for (int k = 0; k < c.n_joint && !bad; ++k) {
        const bool moves = c.src_kind[k] == DEXR_SRC_OPT || c.src_kind[k] == DEXR_SRC_MIMIC;
        if (c.src_kind[k] == DEXR_SRC_OPT) ++n_opt_joints;
        bad = moves ? (c.var[k] < 0 || c.var[k] >= c.n_var) : (c.var[k] != -1);
        if (!bad && c.src_kind[k] == DEXR_SRC_OPT) bad = c.var_joint[c.var[k]] != k || c.vmul[k] != 1.0f;
        if (!bad && c.src_kind[k] == DEXR_SRC_MIMIC) bad = c.var[k] != c.var[c.src_idx[k]] || c.vmul[k] != c.mult[k];
      }
      if (bad || (h.kind != DEXR_KIND_FKONLY && n_opt_joints != c.n_var)) {
        delete m;
        return fail(DEXR_ERR_INVALID, "reduced-variable map of a component is malformed");
      }
      if (c.n_var > m->max_vars) m->max_vars = c.n_var;
    }
    if (c.n_joint > maxj) maxj = c.n_joint;
    if (c.n_frame > m->lds_frames) m->lds_frames = c.n_frame;
    if (c.n_term > m->lds_terms) m->lds_terms = c.n_term;
  }
  m->bucket = pick_bucket(maxj > 0 ? maxj : 1);
  m->max_joints = maxj;
  m->big_nh_rows = maxj * (maxj + 1) / 2;
  for (const dexr_comp_table& c : m->comps)
    for (int k = 0; k < c.n_joint; ++k) {
      m->max_slot = c.save[k] > m->max_slot ? c.save[k] : m->max_slot;
      m->has_mimic = m->has_mimic || c.src_kind[k] == DEXR_SRC_MIMIC;
    }
  default_tuning(m);
  m->wide_ok = build_wide_tables(m);
  select_kernels(m);
  m->tune.lam_jump = family_lam_jump(m, selected_family(m));  // reported value; launches derive it per family
  m->tune.lam_fastdec = family_lam_fastdec(m, selected_family(m));
  m->tune.user_mask = 0u;
  if (m->bucket < 0) {
    delete m;
    return fail(DEXR_ERR_UNSUPPORTED, "component with %d joints exceeds the largest kernel bucket", maxj);
  }
  hipError_t e = hipMalloc((void**)&m->d_comps, m->comps.size() * sizeof(dexr_comp_table));
  if (e == hipSuccess) e = hipMemcpy(m->d_comps, m->comps.data(), m->comps.size() * sizeof(dexr_comp_table), hipMemcpyHostToDevice);
  if (e == hipSuccess && m->wide_ok) {
    e = hipMalloc((void**)&m->d_wide, m->wide_tabs.size() * sizeof(dexr::WideTable));
    if (e == hipSuccess) e = hipMemcpy(m->d_wide, m->wide_tabs.data(), m->wide_tabs.size() * sizeof(dexr::WideTable), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&m->d_queue, (size_t)dexr_model::QSLOTS * h.n_comp * sizeof(unsigned));
  if (e == hipSuccess) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      m->n_cu = cus;
  }
  if (e != hipSuccess) {
    if (m->d_queue) (void)hipFree(m->d_queue);
    if (m->d_comps) (void)hipFree(m->d_comps);
    if (m->d_wide) (void)hipFree(m->d_wide);
    delete m;
    return fail(DEXR_ERR_HIP, "uploading tables failed: %s", hipGetErrorString(e));
  }
  *out = m;
  return DEXR_OK;
}

void dexr_model_destroy(dexr_model* m) {
  if (!m) return;
  if (m->d_comps) (void)hipFree(m->d_comps);
  if (m->d_queue) (void)hipFree(m->d_queue);
  if (m->d_wide) (void)hipFree(m->d_wide);
  if (m->d_gen) (void)hipFree(m->d_gen);
  for (dexr_model::LptSlot& sl : m->lpt) {
    if (sl.buf) (void)hipFree(sl.buf);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  delete m;
}

int dexr_model_info(const dexr_model* m, dexr_model_header* header_out) {
  if (!m || !header_out) return fail(DEXR_ERR_INVALID, "null argument");
  *header_out = m->h;
  return DEXR_OK;
}

int dexr_model_get_tuning(const dexr_model* m, dexr_tuning* out) {
  if (!m || !out) return fail(DEXR_ERR_INVALID, "null argument");
  if (out->struct_size < sizeof(uint32_t) || out->struct_size > sizeof(dexr_tuning))
    return fail(DEXR_ERR_INVALID, "dexr_tuning.struct_size=%u not understood (library: %zu)", out->struct_size, sizeof(dexr_tuning));
  const uint32_t n = out->struct_size;
  std::memcpy(out, &m->tune, n);
  out->struct_size = n;
  return DEXR_OK;
}

int dexr_model_reserve(dexr_model* m, int64_t max_batch) {
  if (!m) return fail(DEXR_ERR_INVALID, "null argument");
  if (max_batch < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  // the hard-frames-first workspaces of the sixteen-lane kernel (launch_wide): every slot, sized for max_batch frames
  if (!m->wide || m->gen) return DEXR_OK;
  const size_t bytes = lpt_slot_bytes(max_batch);
  for (int i = 0; i < dexr_model::LSLOTS; ++i) {
    dexr_model::LptSlot& sl = m->lpt[i];
    if (!sl.done) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    const int rc = lpt_slot_grow(sl, bytes);
    if (rc != DEXR_OK) return rc;
  }
  return DEXR_OK;
}

int dexr_model_set_tuning(dexr_model* m, const dexr_tuning* tuning) {
  if (!m || !tuning) return fail(DEXR_ERR_INVALID, "null argument");
  if (tuning->struct_size < sizeof(uint32_t) || tuning->struct_size > sizeof(dexr_tuning))
    return fail(DEXR_ERR_INVALID, "dexr_tuning.struct_size=%u not understood (library: %zu)", tuning->struct_size, sizeof(dexr_tuning));
  dexr_tuning t = m->tune;  // fields beyond the caller's (older, shorter) struct keep their values
  std::memcpy(&t, tuning, tuning->struct_size);
  t.struct_size = (uint32_t)sizeof(dexr_tuning);
  if (t.kernel < DEXR_KERNEL_AUTO || t.kernel > DEXR_KERNEL_WIDE) return fail(DEXR_ERR_INVALID, "unknown kernel family %d", t.kernel);
  if (t.pivot_rule < -1 || t.pivot_rule > 1) return fail(DEXR_ERR_INVALID, "unknown pivot rule %d", t.pivot_rule);
  if (t.chain < 0 || t.chain > 2) return fail(DEXR_ERR_INVALID, "chain must be 0 (never), 1 (serial-chain kernel + tip pass) or 2 (serial-chain kernel)");
  if (t.longest_first < -1 || t.longest_first > 2) return fail(DEXR_ERR_INVALID, "longest_first must be -1, 0, 1 or 2");
  if (t.fork_streams < -1 || t.fork_streams > 1) return fail(DEXR_ERR_INVALID, "fork_streams must be -1, 0 or 1");
  if (t.sprint_max_batch < -1) return fail(DEXR_ERR_INVALID, "sprint_max_batch must be -1 (policy), 0 (off) or a batch size");
  if (t.sprint_ladder < -1 || t.sprint_ladder > 1) return fail(DEXR_ERR_INVALID, "sprint_ladder must be -1, 0 or 1");
  if (t.tail_passes < -1) return fail(DEXR_ERR_INVALID, "tail_passes must be -1 (policy), 0 (off) or a pass count");
  if (t.persist_from < 0 || t.qchunk < 0 || t.persist_occ < 0 || t.resident_waves < 0 || t.max_blind < 0)
    return fail(DEXR_ERR_INVALID, "negative launch parameter");
  if (!(t.step_cap >= 0) || !(t.lam_jump >= 0) || !(t.lam_fastdec >= 0) || !(t.floor_scale >= 0) || !(t.blind_tol_scale >= 0) || !(t.lam_recover >= 0))
    return fail(DEXR_ERR_INVALID, "negative or non-finite damping parameter");
  // lam_jump / lam_fastdec are caller overrides -- holding for every family -- exactly when the caller says so
  // (user_mask); otherwise each launch uses the default of the family it dispatches (which select_kernels may change now)
  if (tuning->struct_size >= offsetof(dexr_tuning, user_mask) + sizeof(uint32_t)) {
    if (t.user_mask & ~(DEXR_TUNE_LAM_JUMP | DEXR_TUNE_LAM_FASTDEC)) return fail(DEXR_ERR_INVALID, "unknown bits in dexr_tuning.user_mask");
    m->lam_jump_user = (t.user_mask & DEXR_TUNE_LAM_JUMP) ? t.lam_jump : -1.f;
    m->lam_fastdec_user = (t.user_mask & DEXR_TUNE_LAM_FASTDEC) ? t.lam_fastdec : -1.f;
  }
  t.user_mask = (m->lam_jump_user >= 0.f ? DEXR_TUNE_LAM_JUMP : 0u) | (m->lam_fastdec_user >= 0.f ? DEXR_TUNE_LAM_FASTDEC : 0u);
  m->tune = t;
  if (m->gen) return DEXR_OK;  // one kernel serves a generic model
  select_kernels(m);
  m->tune.lam_jump = family_lam_jump(m, selected_family(m));
  m->tune.lam_fastdec = family_lam_fastdec(m, selected_family(m));
  return DEXR_OK;
}

int dexr_model_kernel(const dexr_model* m, int32_t* family, int32_t* bucket, int32_t* chain) {
  if (!m) return fail(DEXR_ERR_INVALID, "null argument");
  if (family && m->gen) *family = DEXR_KERNEL_GENERAL;
  else if (family) *family = m->wide ? DEXR_KERNEL_WIDE : m->red ? DEXR_KERNEL_REDUCED : m->quad ? DEXR_KERNEL_QUAD : m->big ? DEXR_KERNEL_LDS : DEXR_KERNEL_REGISTER;
  if (bucket) *bucket = m->bucket;
  if (chain) *chain = m->tip ? 2 : m->chain ? 1 : 0;
  return DEXR_OK;
}

int dexr_model_lane_plan(const dexr_model* m, int32_t comp, int32_t* n_chain, int32_t* depth, uint8_t* chain_out,
                         uint32_t* anc_rev_out) {
  if (!m) return fail(DEXR_ERR_INVALID, "null argument");
  if (!m->wide_ok) return fail(DEXR_ERR_UNSUPPORTED, "model does not fit the sixteen-lane kernel");
  if (comp < 0 || comp >= (int32_t)m->wide_tabs.size()) return fail(DEXR_ERR_INVALID, "component %d out of range", comp);
  const dexr::WideTable& w = m->wide_tabs[comp];
  if (n_chain) *n_chain = w.n_chain;
  if (depth) *depth = w.depth;
  if (chain_out) std::memcpy(chain_out, w.chain, sizeof(w.chain));
  if (anc_rev_out) std::memcpy(anc_rev_out, w.anc_rev, sizeof(w.anc_rev));
  return DEXR_OK;
}

// Raw-keypoint input needs the rows' keypoint map: target_link_human_indices in the fixed tables, and -- for a model in
// the generic table format -- the validated per-row map (create_generic checks row_ho / row_ht only when has_keypoint_map
// is set: a blob with n_keypoints > 0 but no map would make dexr_gen_kernel index the keypoints with unvalidated rows).
static bool takes_keypoints(const dexr_model* m) { return m->h.n_keypoints > 0 && (!m->gen || m->gen_tab.has_kp); }

static int retarget_dev_impl(const dexr_model* m, int64_t B, const float* ref, bool ref_is_keypoints, const float* fixed,
                             const float* last, uint32_t* state, float* qpos_out, int32_t* status_out,
                             int32_t* iters_out, float* fval_out, const dexr_solve_options* opt, void* stream) {
  if (!m || !ref || !last || !qpos_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (m->h.kind == DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "model is an FK-only table");
  if (ref_is_keypoints && !takes_keypoints(m))
    return fail(DEXR_ERR_INVALID, "model carries no target_link_human_indices: keypoint input not available");
  if (m->h.n_fixed > 0 && !fixed) return fail(DEXR_ERR_INVALID, "model has %d fixed joints but fixed_qpos is NULL", m->h.n_fixed);
  if (B < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  if (B == 0) return DEXR_OK;
  if (opt && opt->precision != 0 && opt->precision != 1) return fail(DEXR_ERR_INVALID, "precision must be 0 (float32) or 1 (float64)");
  const int f64 = (opt && opt->precision == 1) ? 1 : 0;  // float64 arithmetic throughout, float32 result rows
  hipStream_t st = static_cast<hipStream_t>(stream);
  dexr::KernelParams kp;
  fill_params(m, kp, B);
  apply_options(m, kp, opt);
  if (ref_is_keypoints) kp.kpts = ref;
  else kp.ref = ref;
  kp.fixed = fixed;
  kp.last = last;
  kp.state = state;
  kp.qout = qpos_out;
  kp.status = status_out;
  kp.iters = iters_out;
  kp.fval = fval_out;
  if (status_out) HIP_TRY(hipMemsetAsync(status_out, 0, (size_t)B * sizeof(int32_t), st));
  if (iters_out) HIP_TRY(hipMemsetAsync(iters_out, 0, (size_t)B * sizeof(int32_t), st));
  if (fval_out) HIP_TRY(hipMemsetAsync(fval_out, 0, (size_t)B * sizeof(float), st));
  int rc = launch(m, dexr::MODE_SOLVE, f64, kp, st);
  if (rc != DEXR_OK || f64) return rc;
  return polish_launch(m, kp, opt, st);
}

int dexr_retarget_dev(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
                      uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                      const dexr_solve_options* opt, void* stream) {
  return retarget_dev_impl(m, B, ref, false, fixed, last, state, qpos_out, status_out, iters_out, fval_out, opt, stream);
}

int dexr_retarget_kp_dev(const dexr_model* m, int64_t B, const float* keypoints, const float* fixed, const float* last,
                         uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                         const dexr_solve_options* opt, void* stream) {
  return retarget_dev_impl(m, B, keypoints, true, fixed, last, state, qpos_out, status_out, iters_out, fval_out, opt, stream);
}

static int retarget_host(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
                         uint32_t* state, float* q32, double* q64, int32_t* status_out, int32_t* iters_out,
                         float* fval_out, const dexr_solve_options* opt, int f64, bool ref_is_keypoints = false,
                         bool verify_every_step = false) {
  if (!m || !ref || !last || (!q32 && !q64)) return fail(DEXR_ERR_INVALID, "null argument");
  if (m->h.kind == DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "model is an FK-only table");
  if (ref_is_keypoints && !takes_keypoints(m))
    return fail(DEXR_ERR_INVALID, "model carries no target_link_human_indices: keypoint input not available");
  if (m->h.n_fixed > 0 && !fixed) return fail(DEXR_ERR_INVALID, "model has %d fixed joints but fixed_qpos is NULL", m->h.n_fixed);
  if (B < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  if (B == 0) return DEXR_OK;
  const size_t nb = (size_t)B;
  const size_t ref_b = nb * (ref_is_keypoints ? m->h.n_keypoints : m->h.n_ref) * 3 * sizeof(float);
  const size_t fix_b = nb * m->h.n_fixed * sizeof(float);
  const size_t q_b = nb * m->h.n_opt * sizeof(float);
  // one packed block, one copy each way, private stream (dexr_hostctx.hpp)
  std::lock_guard<std::mutex> lock(m->host.mu);
  dexr::HostCtx& hc = m->host;
  dexr::Staging sg;
  const int i_ref = sg.add(dexr::Staging::IN, ref, nullptr, ref_b);
  const int i_fix = sg.add(dexr::Staging::IN, fixed, nullptr, fix_b);
  const int i_last = sg.add(dexr::Staging::IN, last, nullptr, q_b);
  const int i_state = sg.add(dexr::Staging::INOUT, state, state, nb * sizeof(uint32_t));
  const int i_status = sg.add(dexr::Staging::INOUT, nullptr, status_out, nb * sizeof(int32_t));  // zeroed: the
  const int i_iters = sg.add(dexr::Staging::INOUT, nullptr, iters_out, nb * sizeof(int32_t));    // kernels combine
  const int i_fval = sg.add(dexr::Staging::INOUT, nullptr, fval_out, nb * sizeof(float));        // components atomically
  const int i_q = sg.add(dexr::Staging::OUT, nullptr, q32, q_b);
  const int i_q64 = sg.add(dexr::Staging::OUT, nullptr, q64, q64 ? nb * m->h.n_opt * sizeof(double) : 0);
  HIP_TRY(sg.upload(hc));
  dexr::KernelParams kp;
  fill_params(m, kp, B);
  apply_options(m, kp, opt);
  if (ref_is_keypoints) kp.kpts = sg.dev<float>(hc, i_ref);
  else kp.ref = sg.dev<float>(hc, i_ref);
  kp.fixed = sg.dev<float>(hc, i_fix);
  kp.last = sg.dev<float>(hc, i_last);
  kp.state = sg.dev<uint32_t>(hc, i_state);
  kp.qout = sg.dev<float>(hc, i_q);
  kp.qout64 = q64 ? sg.dev<double>(hc, i_q64) : nullptr;
  kp.status = sg.dev<int32_t>(hc, i_status);
  kp.iters = sg.dev<int32_t>(hc, i_iters);
  kp.fval = sg.dev<float>(hc, i_fval);
  if (verify_every_step) kp.blind_tol = 0.f;  // dexr_retarget_f64 is the validation path: every step it reports has been verified
  int rc = launch(m, dexr::MODE_SOLVE, f64, kp, hc.st);
  if (rc != DEXR_OK) return rc;
  if (!f64) {
    rc = polish_launch(m, kp, opt, hc.st);
    if (rc != DEXR_OK) return rc;
  }
  HIP_TRY(sg.download(hc));
  return DEXR_OK;
}

int dexr_retarget(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
                  uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                  const dexr_solve_options* opt) {
  if (opt && opt->precision != 0 && opt->precision != 1) return fail(DEXR_ERR_INVALID, "precision must be 0 (float32) or 1 (float64)");
  return retarget_host(m, B, ref, fixed, last, state, qpos_out, nullptr, status_out, iters_out, fval_out, opt,
                       (opt && opt->precision == 1) ? 1 : 0);
}

int dexr_retarget_kp(const dexr_model* m, int64_t B, const float* keypoints, const float* fixed, const float* last,
                     uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                     const dexr_solve_options* opt) {
  if (opt && opt->precision != 0 && opt->precision != 1) return fail(DEXR_ERR_INVALID, "precision must be 0 (float32) or 1 (float64)");
  return retarget_host(m, B, keypoints, fixed, last, state, qpos_out, nullptr, status_out, iters_out, fval_out, opt,
                       (opt && opt->precision == 1) ? 1 : 0, true);
}

int dexr_retarget_f64(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
                      uint32_t* state, double* qpos_out, int32_t* status_out, int32_t* iters_out,
                      const dexr_solve_options* opt) {
  return retarget_host(m, B, ref, fixed, last, state, nullptr, qpos_out, status_out, iters_out, nullptr, opt, 1, false, true);
}

int dexr_eval(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
              const double* x, uint32_t* state, double* f_out, double* grad_out) {
  if (!m || !ref || !last || !x || !f_out || !grad_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (m->h.kind == DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "model is an FK-only table");
  if (m->h.n_fixed > 0 && !fixed) return fail(DEXR_ERR_INVALID, "model has %d fixed joints but fixed_qpos is NULL", m->h.n_fixed);
  if (B < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  if (B == 0) return DEXR_OK;
  const size_t nb = (size_t)B;
  const size_t ref_b = nb * m->h.n_ref * 3 * sizeof(float), fix_b = nb * m->h.n_fixed * sizeof(float);
  const size_t q_b = nb * m->h.n_opt * sizeof(float), x_b = nb * m->h.n_opt * sizeof(double);
  std::lock_guard<std::mutex> lock(m->host.mu);
  dexr::HostCtx& hc = m->host;
  dexr::Staging sg;
  const int i_ref = sg.add(dexr::Staging::IN, ref, nullptr, ref_b);
  const int i_fix = sg.add(dexr::Staging::IN, fixed, nullptr, fix_b);
  const int i_last = sg.add(dexr::Staging::IN, last, nullptr, q_b);
  const int i_x = sg.add(dexr::Staging::IN, x, nullptr, x_b);
  const int i_state = sg.add(dexr::Staging::INOUT, state, state, nb * sizeof(uint32_t));
  const int i_f = sg.add(dexr::Staging::INOUT, nullptr, f_out, nb * sizeof(double));  // zeroed: summed over components
  const int i_g = sg.add(dexr::Staging::INOUT, nullptr, grad_out, x_b);
  HIP_TRY(sg.upload(hc));
  dexr::KernelParams kp;
  fill_params(m, kp, B);
  kp.ref = sg.dev<float>(hc, i_ref);
  kp.fixed = sg.dev<float>(hc, i_fix);
  kp.last = sg.dev<float>(hc, i_last);
  kp.xin = sg.dev<double>(hc, i_x);
  kp.state = sg.dev<uint32_t>(hc, i_state);
  kp.f64out = sg.dev<double>(hc, i_f);
  kp.g64out = sg.dev<double>(hc, i_g);
  int rc = launch(m, dexr::MODE_EVAL, 1, kp, hc.st);
  if (rc != DEXR_OK) return rc;
  HIP_TRY(sg.download(hc));
  return DEXR_OK;
}

int dexr_fk(const dexr_model* m, int64_t B, const double* q, double* pos_out) {
  if (!m || !q || !pos_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (m->h.kind != DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "model is not an FK-only table");
  if (B < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  if (B == 0) return DEXR_OK;
  const size_t nb = (size_t)B;
  const size_t q_b = nb * m->h.n_q * sizeof(double), p_b = nb * m->h.n_ref * 3 * sizeof(double);
  std::lock_guard<std::mutex> lock(m->host.mu);
  dexr::HostCtx& hc = m->host;
  dexr::Staging sg;
  const int i_q = sg.add(dexr::Staging::IN, q, nullptr, q_b);
  const int i_p = sg.add(dexr::Staging::INOUT, nullptr, pos_out, p_b);
  HIP_TRY(sg.upload(hc));
  dexr::KernelParams kp;
  fill_params(m, kp, B);
  kp.xin = sg.dev<double>(hc, i_q);
  kp.f64out = sg.dev<double>(hc, i_p);
  int rc = launch(m, dexr::MODE_FK, 1, kp, hc.st);
  if (rc != DEXR_OK) return rc;
  HIP_TRY(sg.download(hc));
  return DEXR_OK;
}

int dexr_retarget_seq_dev(const dexr_model* m, int64_t B, int32_t T, const float* inputs, int32_t inputs_are_keypoints,
                          const float* fixed, float* last_inout, uint32_t* state_inout, float* qpos_raw_out,
                          int32_t* status_out, float joint_limit_eps, const dexr_solve_options* opt, void* stream) {
  if (!m || !inputs || !last_inout || !qpos_raw_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (m->h.kind == DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "model is an FK-only table");
  if (inputs_are_keypoints && !takes_keypoints(m))
    return fail(DEXR_ERR_INVALID, "model carries no target_link_human_indices: keypoint input not available");
  if (m->h.n_fixed > 0 && !fixed) return fail(DEXR_ERR_INVALID, "model has %d fixed joints but fixed_qpos is NULL", m->h.n_fixed);
  if (B < 0 || T < 0) return fail(DEXR_ERR_INVALID, "negative batch or sequence length");
  if (!(joint_limit_eps >= 0.f)) return fail(DEXR_ERR_INVALID, "joint_limit_eps must be >= 0");
  if (B == 0 || T == 0) return DEXR_OK;
  if ((int64_t)T * B > 0x7fffffffLL) return fail(DEXR_ERR_INVALID, "T x B too large for one launch");
  if (opt && opt->precision != 0 && opt->precision != 1) return fail(DEXR_ERR_INVALID, "precision must be 0 (float32) or 1 (float64)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  dexr::KernelParams kp;
  fill_params(m, kp, B);
  apply_options(m, kp, opt);
  if (inputs_are_keypoints) kp.kpts = inputs;
  else kp.ref = inputs;
  kp.fixed = fixed;
  kp.last = last_inout;
  kp.state = state_inout;
  kp.qout = qpos_raw_out;
  kp.status = status_out;
  kp.T = T;
  kp.seq_stride = B;
  kp.clip_eps = joint_limit_eps;
  if (status_out) HIP_TRY(hipMemsetAsync(status_out, 0, (size_t)T * (size_t)B * sizeof(int32_t), st));
  // a polish launch cannot be interleaved with the carry: models that need it solve in float64 throughout
  const int f64 = ((opt && opt->precision == 1) || polish_wanted(m, opt)) ? 1 : 0;
  int rc = launch(m, dexr::MODE_SOLVE, f64, kp, st);
  if (rc != DEXR_OK) return rc;
  // SeqRetargeting.last_qpos after the last frame = its raw answer (seq_retarget.py:124)
  HIP_TRY(hipMemcpyAsync(last_inout, qpos_raw_out + (size_t)(T - 1) * (size_t)B * m->h.n_opt,
                         (size_t)B * m->h.n_opt * sizeof(float), hipMemcpyDeviceToDevice, st));
  return DEXR_OK;
}

int dexr_seq_compose_dev(int64_t B, int32_t T, int32_t n_q, int32_t n_opt, int32_t n_fixed, const int32_t* dof_kind,
                         const int32_t* dof_idx, const double* dof_mult, const double* dof_off, const float* qpos_raw,
                         const float* fixed, double alpha, double* filter_inout, int32_t first_frame_initialises,
                         double* robot_qpos_out, void* stream) {
  if (!dof_kind || !dof_idx || !dof_mult || !dof_off || !qpos_raw || !robot_qpos_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (n_q < 1 || n_q > DEXR_MAX_DOF) return fail(DEXR_ERR_INVALID, "n_q=%d outside 1..%d", n_q, DEXR_MAX_DOF);
  if (B < 0 || T < 0 || n_opt < 0 || n_fixed < 0) return fail(DEXR_ERR_INVALID, "negative size");
  const bool use_filter = alpha >= 0.0 && alpha <= 1.0;
  if (use_filter && !filter_inout) return fail(DEXR_ERR_INVALID, "a low-pass coefficient was given but filter_inout is NULL");
  for (int j = 0; j < n_q; ++j) {
    const int k = dof_kind[j], i = dof_idx[j];
    const bool bad = k < 0 || k > 2 || i < 0 || (k == 0 && i >= n_opt) || (k == 1 && i >= n_fixed) ||
                     (k == 2 && (i >= n_q || dof_kind[i] == 2));
    if (bad) return fail(DEXR_ERR_INVALID, "dof %d: malformed source (kind %d, idx %d)", j, k, i);
    if (k == 1 && !fixed) return fail(DEXR_ERR_INVALID, "dof %d is a fixed joint but fixed is NULL", j);
  }
  if (B == 0 || T == 0) return DEXR_OK;
  const hipError_t e = dexr_seq_compose_launch(B, T, n_q, n_opt, n_fixed, dof_kind, dof_idx, dof_mult, dof_off, qpos_raw, fixed,
                                               alpha, use_filter ? 1 : 0, first_frame_initialises ? 1 : 0, filter_inout,
                                               robot_qpos_out, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "compose kernel launch failed: %s", hipGetErrorString(e));
  return DEXR_OK;
}

namespace {
// Fork / join streams for entry points that enqueue INDEPENDENT launches (one per model of a fleet batch): the launches go
// to internal streams ordered after the caller's stream by one event and the caller's stream is ordered after all of them
// before the call returns, so the call keeps plain stream semantics while the launches' tails overlap.  Legal inside a
// stream capture (the internal streams join the capture through the fork event).  One pool per device, lazily created.
struct ForkPool {
  std::mutex mu;
  hipStream_t aux[DEXR_FLEET_MAX_MODELS] = {};
  hipEvent_t join[DEXR_FLEET_MAX_MODELS] = {};
  hipEvent_t fork = nullptr;
};
ForkPool* fork_pool() {
  static ForkPool* pools[64] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!pools[dev]) pools[dev] = new ForkPool();
  return pools[dev];
}
}  // namespace

// (the bucketing workspace + the index list + what ordering ONE DexPilot model's bucket hard-frames-first needs)
size_t dexr_fleet_workspace_bytes(int64_t B) {
  return (dexr_fleet_ws_ints() + (size_t)(B > 0 ? B : 0) + dexr_fleet_order_ws_ints(B)) * sizeof(int32_t);
}

int dexr_retarget_multi_dev(const dexr_model* const* models, int32_t n_models, int64_t B, const int32_t* model_id,
                            const float* keypoints, const float* fixed, int32_t ld_fixed, const float* last, int32_t ld,
                            uint32_t* state, float* qpos_out, int32_t* status_out, const dexr_solve_options* opt,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!models || !model_id || !keypoints || !last || !qpos_out || !workspace) return fail(DEXR_ERR_INVALID, "null argument");
  if (n_models < 1 || n_models > DEXR_FLEET_MAX_MODELS) return fail(DEXR_ERR_INVALID, "n_models=%d outside 1..%d", n_models, DEXR_FLEET_MAX_MODELS);
  if (B < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  if (B > 0x7fffffffLL) return fail(DEXR_ERR_INVALID, "batch too large for one call");
  if (workspace_bytes < dexr_fleet_workspace_bytes(B)) return fail(DEXR_ERR_INVALID, "workspace of %zu B, %zu B needed", workspace_bytes, dexr_fleet_workspace_bytes(B));
  if (opt && opt->precision != 0) return fail(DEXR_ERR_INVALID, "fleet batches run the float32 / mixed-precision kernels");
  for (int i = 0; i < n_models; ++i) {
    const dexr_model* m = models[i];
    if (!m) return fail(DEXR_ERR_INVALID, "models[%d] is NULL", i);
    if (m->h.kind == DEXR_KIND_FKONLY) return fail(DEXR_ERR_INVALID, "models[%d] is an FK-only table", i);
    if (!takes_keypoints(m)) return fail(DEXR_ERR_INVALID, "models[%d] carries no target_link_human_indices", i);
    if (m->h.n_fixed > 0 && (!fixed || m->h.n_fixed > ld_fixed))
      return fail(DEXR_ERR_INVALID, "models[%d] has %d caller-supplied fixed joints but fixed rows are %d long", i, m->h.n_fixed, fixed ? ld_fixed : 0);
    if (m->h.n_opt > ld) return fail(DEXR_ERR_INVALID, "models[%d] optimises %d joints but rows are %d long", i, m->h.n_opt, ld);
    if (m->h.kind == DEXR_KIND_DEXPILOT && !state) return fail(DEXR_ERR_INVALID, "models[%d] is a DexPilot model but state is NULL", i);
  }
  if (B == 0) return DEXR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t* ws = static_cast<int32_t*>(workspace);
  hipError_t e = dexr_fleet_bucket_launch(n_models, B, model_id, ws, st);
  if (e != hipSuccess) return fail(DEXR_ERR_HIP, "fleet bucketing launch failed: %s", hipGetErrorString(e));
  if (status_out) HIP_TRY(hipMemsetAsync(status_out, 0, (size_t)B * sizeof(int32_t), st));
  const int32_t* bucket = ws + 2 * DEXR_FLEET_MAX_MODELS;
  const int32_t* perm = ws + dexr_fleet_ws_ints();
  // The models' buckets are disjoint rows: their launches are independent.  With fork_streams on, the first HEAVY model
  // (components of 9+ joints: a persistent kernel that holds every SIMD's register file and ends in a long tail of a few
  // slow frames) stays on the caller's stream, so it starts the moment the bucketing kernels end; further heavy models
  // get an internal stream each; ALL light models (small components, tens of microseconds each) go to one more internal
  // stream behind the fork event -- they reach the GPU a few microseconds later, find it full, and fill the CUs as the
  // heavy launch's waves retire.  (Measured the other way round -- light models on the caller's stream, heavy ones
  // forked -- the light kernels usually won the race for the CUs and the persistent blocks that had to wait stretched the
  // heavy launch from 1.0 to 1.4 ms.)
  ForkPool* fp = (n_models > 1 && models[0]->tune.fork_streams != 0) ? fork_pool() : nullptr;
  std::unique_lock<std::mutex> lock;
  if (fp) {
    lock = std::unique_lock<std::mutex>(fp->mu);
    // every stream and event the fork may need exists BEFORE the fork event is recorded: an allocation failure returns
    // here, with nothing forked (an early return after work has been enqueued on an internal stream would leave the
    // caller's stream un-joined -- buffers in use, an unjoined fork inside a stream capture)
    if (!fp->fork) HIP_TRY(hipEventCreateWithFlags(&fp->fork, hipEventDisableTiming));
    for (int sl = 0; sl < DEXR_FLEET_MAX_MODELS; ++sl) {
      if (!fp->aux[sl]) HIP_TRY(hipStreamCreateWithFlags(&fp->aux[sl], hipStreamNonBlocking));
      if (!fp->join[sl]) HIP_TRY(hipEventCreateWithFlags(&fp->join[sl], hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(fp->fork, st));
  }
  int rc_all = DEXR_OK;
  bool forked[DEXR_FLEET_MAX_MODELS] = {};
  int n_heavy = 0;
  for (int i = 0; i < n_models; ++i) n_heavy += selected_family(models[i]) != FAM_REGISTER || models[i]->gen;
  const int light_slot = 0;  // stream slot 0 of the pool serves the light models (model 0's own slot is free: see below)
  bool first_heavy_placed = false, ordered_one = false;
  for (int pass = 0; pass < 2 && rc_all == DEXR_OK; ++pass) {
    for (int i = 0; i < n_models; ++i) {
      const dexr_model* m = models[i];
      const bool heavy = selected_family(m) != FAM_REGISTER || m->gen;
      if (heavy != (pass == 0)) continue;
      hipStream_t si = st;
      int slot = -1;
      if (fp && n_heavy > 0) {
        if (heavy && first_heavy_placed) slot = 1 + (i % (DEXR_FLEET_MAX_MODELS - 1));
        else if (!heavy) slot = light_slot;
      }
      if (heavy) first_heavy_placed = true;
      if (slot >= 0) {
        si = fp->aux[slot];
        if (!forked[slot]) {
          const hipError_t we = hipStreamWaitEvent(si, fp->fork, 0);
          if (we != hipSuccess) {  // join what has been forked so far, then report
            rc_all = fail(DEXR_ERR_HIP, "hipStreamWaitEvent(fork) failed: %s", hipGetErrorString(we));
            break;
          }
        }
        forked[slot] = true;
      }
      dexr::KernelParams kp;
      fill_params(m, kp, B);  // B: upper bound of the bucket size (launch geometry); the kernel reads the real count
      apply_options(m, kp, opt);
      kp.kpts = keypoints;
      kp.fixed = fixed;
      kp.ldf = ld_fixed;
      kp.last = last;
      kp.state = m->h.kind == DEXR_KIND_DEXPILOT ? state : nullptr;
      kp.qout = qpos_out;
      kp.status = status_out;
      kp.ld = ld;
      kp.perm = perm;
      kp.bucket = bucket + 2 * i;
      int rc = DEXR_OK;
      // The first heavy DexPilot model's bucket is walked hard frames first (keys from the projection state, like
      // launch_wide's plain batches): its launch is the step's critical path and ends in the tail of exactly those frames.
      // Gate: the host never learns the bucket sizes, so the model's share is ESTIMATED as an even split B / n_models
      // -- a ~10 000-frame bucket of a 40 000-frame fleet does not pay ~25 us of key / bucketing / gather kernels for a launch
      // that is not tail-bound; and only the state-key modes ask for it (longest_first -1: automatic, 2: always; 1 is the
      // F(x0) screening mode of plain batches, 0 is off).
      const int lf = m->tune.longest_first;
      if (!ordered_one && heavy && !m->gen && selected_family(m) == FAM_WIDE && m->h.kind == DEXR_KIND_DEXPILOT && m->h.n_opt >= 9 &&
          (lf == 2 || (lf < 0 && B / n_models >= 4 * (int64_t)m->n_cu * 4 * 2 * 4))) {
        ordered_one = true;
        const int32_t* operm = nullptr;
        const int32_t* oseg = nullptr;
        const hipError_t oe = dexr_fleet_dexpilot_order_launch(B, keypoints, state, perm, bucket + 2 * i, kp.n_kp, kp.h_task, kp.h_origin,
                                                               kp.num_fingers, kp.project_dist, kp.escape_dist,
                                                               ws + dexr_fleet_ws_ints() + B, &operm, &oseg, si);
        if (oe != hipSuccess) rc = fail(DEXR_ERR_HIP, "fleet ordering kernels failed: %s", hipGetErrorString(oe));
        kp.perm = operm;
        kp.bucket = oseg;
      }
      if (rc == DEXR_OK) rc = launch(m, dexr::MODE_SOLVE, 0, kp, si);
      if (rc == DEXR_OK) rc = polish_launch(m, kp, opt, si);
      if (rc != DEXR_OK) {  // still join what has been forked, then report
        rc_all = rc;
        break;
      }
    }
  }
  std::string err = g_err;
  for (int sl = 0; sl < DEXR_FLEET_MAX_MODELS; ++sl) {  // every forked stream is joined, whatever failed before
    if (!forked[sl]) continue;
    hipError_t je = hipEventRecord(fp->join[sl], fp->aux[sl]);
    if (je == hipSuccess) je = hipStreamWaitEvent(st, fp->join[sl], 0);
    if (je != hipSuccess && rc_all == DEXR_OK) {
      rc_all = fail(DEXR_ERR_HIP, "joining internal stream %d failed: %s", sl, hipGetErrorString(je));
      err = g_err;
    }
  }
  if (rc_all != DEXR_OK) g_err = err;
  return rc_all;
}

int dexr_retarget_multi(const dexr_model* const* models, int32_t n_models, int64_t B, const int32_t* model_id,
                        const float* keypoints, const float* fixed, int32_t ld_fixed, const float* last, int32_t ld,
                        uint32_t* state, float* qpos_out, int32_t* status_out, const dexr_solve_options* opt) {
  if (!models || !model_id || !keypoints || !last || !qpos_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (ld_fixed < 0 || (fixed && ld_fixed < 1)) return fail(DEXR_ERR_INVALID, "fixed rows of length %d", ld_fixed);
  if (n_models < 1 || n_models > DEXR_FLEET_MAX_MODELS) return fail(DEXR_ERR_INVALID, "n_models=%d outside 1..%d", n_models, DEXR_FLEET_MAX_MODELS);
  if (!models[0]) return fail(DEXR_ERR_INVALID, "models[0] is NULL");
  if (B < 0 || ld < 1) return fail(DEXR_ERR_INVALID, "negative batch or row length");
  if (B == 0) return DEXR_OK;
  const size_t nb = (size_t)B;
  // staged through models[0]'s host context (pinned / device block, the library stream); rows the call leaves untouched
  // (unknown model ids, columns beyond a model's n_opt) must come back as the caller passed them: qpos_out is in-out
  std::lock_guard<std::mutex> lock(models[0]->host.mu);
  dexr::HostCtx& hc = models[0]->host;
  dexr::Staging sg;
  const size_t ws_b = dexr_fleet_workspace_bytes(B);
  const int i_id = sg.add(dexr::Staging::IN, model_id, nullptr, nb * sizeof(int32_t));
  const int i_kp = sg.add(dexr::Staging::IN, keypoints, nullptr, nb * 21 * 3 * sizeof(float));
  const int i_fix = sg.add(dexr::Staging::IN, fixed, nullptr, fixed ? nb * (size_t)ld_fixed * sizeof(float) : 0);
  const int i_last = sg.add(dexr::Staging::IN, last, nullptr, nb * (size_t)ld * sizeof(float));
  const int i_state = sg.add(dexr::Staging::INOUT, state, state, nb * sizeof(uint32_t));
  const int i_q = sg.add(dexr::Staging::INOUT, qpos_out, qpos_out, nb * (size_t)ld * sizeof(float));
  const int i_status = sg.add(dexr::Staging::INOUT, nullptr, status_out, nb * sizeof(int32_t));
  const int i_ws = sg.add(dexr::Staging::OUT, nullptr, nullptr, ws_b);
  HIP_TRY(sg.upload(hc));
  const int rc = dexr_retarget_multi_dev(models, n_models, B, sg.dev<int32_t>(hc, i_id), sg.dev<float>(hc, i_kp),
                                         fixed ? sg.dev<float>(hc, i_fix) : nullptr, ld_fixed, sg.dev<float>(hc, i_last), ld, sg.dev<uint32_t>(hc, i_state), sg.dev<float>(hc, i_q),
                                         status_out ? sg.dev<int32_t>(hc, i_status) : nullptr, opt, sg.dev<void>(hc, i_ws), ws_b, hc.st);
  if (rc != DEXR_OK) return rc;
  HIP_TRY(sg.download(hc));
  return DEXR_OK;
}

int dexr_mano_keypoints_dev(int64_t B, const float* keypoints, const float* operator2mano, float* joint_pos_out,
                            float* wrist_rot_out, void* stream) {
  if (!keypoints || !operator2mano || !joint_pos_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (B < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  if (B == 0) return DEXR_OK;
  const int rc = dexr_prep_launch(B, keypoints, operator2mano, joint_pos_out, wrist_rot_out, static_cast<hipStream_t>(stream));
  if (rc != 0) return fail(DEXR_ERR_HIP, "keypoint kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  return DEXR_OK;
}

int dexr_mano_keypoints(int64_t B, const float* keypoints, const float* operator2mano, float* joint_pos_out,
                        float* wrist_rot_out) {
  if (!keypoints || !operator2mano || !joint_pos_out) return fail(DEXR_ERR_INVALID, "null argument");
  if (B < 0) return fail(DEXR_ERR_INVALID, "negative batch");
  if (B == 0) return DEXR_OK;
  const size_t kp_b = (size_t)B * 21 * 3 * sizeof(float), r_b = (size_t)B * 9 * sizeof(float);
  // no model handle on this entry point: one process-wide staging context (never destroyed: the HIP runtime may be
  // gone by the time static destructors run)
  static dexr::HostCtx& ctx = *new dexr::HostCtx();
  std::lock_guard<std::mutex> lock(ctx.mu);
  dexr::Staging sg;
  const int i_in = sg.add(dexr::Staging::IN, keypoints, nullptr, kp_b);
  const int i_out = sg.add(dexr::Staging::OUT, nullptr, joint_pos_out, kp_b);
  const int i_rot = sg.add(dexr::Staging::OUT, nullptr, wrist_rot_out, wrist_rot_out ? r_b : 0);
  HIP_TRY(sg.upload(ctx));
  const int rc = dexr_mano_keypoints_dev(B, sg.dev<float>(ctx, i_in), operator2mano, sg.dev<float>(ctx, i_out),
                                         wrist_rot_out ? sg.dev<float>(ctx, i_rot) : nullptr, ctx.st);
  if (rc != DEXR_OK) return rc;
  HIP_TRY(sg.download(ctx));
  return DEXR_OK;
}

}  // extern "C"
