/* dexr.h -- C ABI of libdexr.so: the MI355X (gfx950) batched retargeting solver.
 *
 * The reference (dexsuite/dex-retargeting v0.5.0) has no FFI layer: its boundary for this path is the
 * Python method  Optimizer.retarget(ref_value, fixed_qpos, last_qpos) -> float32 qpos
 * (/root/reference/src/dex_retargeting/optimizer.py:77-102) plus the nlopt-style closure
 * objective(x, grad) -> float (optimizer.py:146, 249, 510) and RobotWrapper's forward kinematics
 * (/root/reference/src/dex_retargeting/robot_wrapper.py:82-87).  The entry points below are what a
 * ctypes binding on the reference side binds instead of nlopt + pinocchio + torch.
 *
 * Conventions: plain pointers and sizes only; all arrays C-contiguous; every function returns 0 on success
 * or a negative error code (message via dexr_last_error(), thread-local); no exceptions cross the ABI.
 * "_dev" entry points take DEVICE pointers and enqueue on `stream` (a hipStream_t passed as void*, NULL =
 * default stream) without synchronising; the others take HOST pointers, copy, run and synchronise.
 */
#ifndef DEXR_H
#define DEXR_H

#include <stddef.h>
#include <stdint.h>

#include "dexr_tables.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dexr_model dexr_model; /* opaque: compiled tables resident in HBM */

/* error codes */
#define DEXR_OK 0
#define DEXR_ERR_INVALID (-1) /* bad argument / malformed table blob   */
#define DEXR_ERR_HIP (-2)     /* HIP runtime error (no device, OOM, ..) */
#define DEXR_ERR_UNSUPPORTED (-3)

/* per-item solver status written to status_out (max over the item's components) */
#define DEXR_STATUS_CONVERGED 0
#define DEXR_STATUS_MAXITER 1
#define DEXR_STATUS_FALLBACK 2 /* non-finite state: last_qpos returned, like optimizer.py:100-102 */

/* solver options; zero-initialise then call dexr_default_options() */
typedef struct dexr_solve_options {
  int32_t max_iter;   /* outer iterations (default 64)                                    */
  float tol;          /* stop when the accepted step's inf-norm < tol [rad|m] (default 2e-6) */
  float lambda0;      /* initial Levenberg-Marquardt damping (default 1e-4)                */
  int32_t newton;     /* 1: add the second-order kinematic term to the Hessian (default 1) */
  int32_t precision;  /* 0: float32 / mixed-precision kernels (default); 1: float64 arithmetic throughout (the
                         reference's own arithmetic type; register kernel, no polish pass).  Device-pointer entry
                         points accept both; the result rows are float32 either way (optimizer.py:99)            */
  int32_t polish;     /* float64 polishing iterations run after the float32 solve, started at its answer:
                         -1 auto (default: up to 24 for position / DexPilot models, whose float32 rounding floor sits
                         near 1e-4 rad; 0 for vector models), 0 off, n > 0 at most n iterations           */
  int32_t strict;     /* float64 polish after the mixed-precision kernels (float64 kinematics and value, float32
                         gradient and Hessian) that serve large components.  0 (default): only after the quad / LDS
                         kernels on models with mimic joints; 1: after every mixed-precision kernel; -1: never.      */
} dexr_solve_options;

/* Per-model launch / damping parameters.  These are the values the launcher derives from the model's shape and from
 * measurements on MI355X; they are exposed so that a deployment (or the profiling tools under
 * tools/) can override them for ONE model handle, explicitly, instead of through process-wide environment variables.
 * Nothing in the library reads the environment.  Obtain the current values with dexr_model_get_tuning(), change
 * fields, hand them back with dexr_model_set_tuning() (not thread-safe against launches in flight on that handle). */
#define DEXR_KERNEL_AUTO (-1)
#define DEXR_KERNEL_REGISTER 0 /* one lane per (frame, component), Hessian in registers (dexr_kernel.hpp)            */
#define DEXR_KERNEL_QUAD 1     /* four lanes per frame, distributed Hessian rows (dexr_quad.hpp)                      */
#define DEXR_KERNEL_LDS 2      /* one lane per frame, Hessian in LDS (dexr_big.hpp)                                   */
#define DEXR_KERNEL_REDUCED 3  /* one lane per frame, Hessian of the optimised VARIABLES (mimic joints folded while the
                                  Jacobian is formed) in registers, kinematics in LDS (dexr_red.hpp)                   */
#define DEXR_KERNEL_WIDE 4     /* sixteen lanes per frame: chain-parallel kinematics, 4 x 4 lane grid for the Hessian and
                                  its Cholesky factor, no re-assembly after a rejected step (dexr_wide.hpp)             */
#define DEXR_KERNEL_GENERAL 5  /* reported by dexr_model_kernel for models in the generic table format (dexr_tables.h): one
                                  wavefront per frame, every table in memory, float64 (dexr_gen.hpp); not selectable      */
typedef struct dexr_tuning {
  uint32_t struct_size; /* sizeof(dexr_tuning) of the caller's header: lets the struct grow compatibly          */
  int32_t kernel;       /* DEXR_KERNEL_*: float32 solve kernel family (AUTO: measured policy, dexr_api.hip).  Under AUTO
                           the family may follow the SIZE of the call: models the policy gives to the reduced-variable kernel
                           (mimic vector models) take the sixteen-lane kernel for batches of <= 16 384 frames (2 x faster there;
                           dexr_model_kernel reports the large-batch family).  An explicit family holds at every size.         */
  int32_t chain;        /* 1: serial-chain specialisation (+ its tip pass) where the tables allow it (default), 2: serial-chain
                           specialisation without the tip pass, 0: never                                           */
  int32_t persist_occ;  /* small components: resident waves per SIMD in queue mode (0: derived from the kernel)  */
  int32_t persist_from; /* small components: queue mode from this many 64-frame tiles per resident wave (8)       */
  int32_t qchunk;       /* frames a wave takes from the queue per atomic (256)                                    */
  int32_t resident_waves; /* quad / LDS kernels: resident waves (0: one per SIMD resp. what the LDS allows)       */
  int32_t max_blind;    /* accepted steps below the rounding floor of F before the solve stops (8)                */
  int32_t stall_from;   /* see dexr_kernel.hpp "stalled"                                                          */
  float stall_ratio, stall_cap;
  float lam_jump;       /* rejected step: lambda >= lam_jump x curvature scale (0: plain Nielsen)                 */
  float lam_fastdec;    /* accepted step with rho > 0.9: lambda *= lam_fastdec (0: Nielsen's 1/3)                 */
  float floor_scale;    /* mixed-precision kernels: value differences below floor_scale x |F| are unverifiable    */
  float step_cap;       /* trust radius per joint [rad|m] (0: off)                                                */
  float blind_tol_scale; /* a verified undamped Newton step shorter than blind_tol_scale x tol ends the solve (10) */
  int32_t pivot_rule;   /* sixteen-lane kernel on the variable grid (models with mimic joints): 0 plain Cholesky (a non-positive pivot fails the pass; lam_jump scales
                           the Rayleigh quotient of the failed step), 1 modified Cholesky (the pivot is reflected, the
                           step judged by the decrease and stretched to the trust radius; lam_jump scales mean diag H),
                           -1 measured policy (1 for DexPilot models with mimic joints)                             */
  int32_t longest_first; /* sixteen-lane kernel, plain batches: hard frames first (a launch is otherwise bound by slow frames
                           the queue hands out late).  2: DexPilot models, keys from the projection state -- a projection
                           bit changes in this frame / a projection is active / neither -- one elementwise kernel, no
                           screening launch; 1: a screening launch evaluates F at the start points, frames above 1.3 x the
                           batch mean are solved first (costs more than it gains, see dexr_api.hip launch_wide); 0 off;
                           -1 measured policy: the state keys for DexPilot batches of >= 32 768 frames, else off      */
  float lam_recover;    /* small components: accepted step with rho > 0.9 while lambda > 10 lambda0 and at most two steps of
                           the solve were rejected (the damping a rejection raised is being taken back): lambda *=
                           lam_recover.  0 (default): lam_fastdec there too.  Measured at 0.003 (65 536 tracking frames):
                           Allegro vector 61.5 -> 56 us, Inspire 67 -> 64, Ability 87 -> 84, but LEAP vector 54 -> 62: the
                           frames with >= 8 passes drop everywhere (LEAP: 108 -> 30), the single slowest frame -- which
                           sets a launch's duration -- moves either way (LEAP: 11 -> 14 passes), hence opt-in          */
  int32_t fork_streams; /* dexr_retarget_multi_dev (read from models[0]): the first model whose components have 9+ joints
                           stays on `stream`; the small-component models are enqueued on an internal stream ordered after
                           `stream` by an event and joined before the call returns, so that they fill the CUs the big
                           launch's tail leaves idle (further big models get internal streams of their own).  1 on, 0 off
                           (everything on `stream`), -1 policy (on)                                                     */
  uint32_t user_mask;   /* which damping fields are CALLER OVERRIDES: DEXR_TUNE_LAM_JUMP, DEXR_TUNE_LAM_FASTDEC.  Their
                           defaults depend on the kernel family a launch dispatches (they scale different quantities per
                           family), so dexr_model_get_tuning reports the value in use for the selected family with the bit
                           CLEAR, and dexr_model_set_tuning takes lam_jump / lam_fastdec as an override -- which then holds
                           for every family -- only when the bit is SET; with the bit clear the field is ignored and any
                           earlier override is dropped.  (Until round 3 an override was inferred from "differs from the
                           reported value", which pinned stale values of re-used structs and made defaults sticky.)
                           A caller whose (older) struct ends before this field leaves the override state untouched.   */
  int32_t sprint_max_batch; /* sixteen-lane kernel, plain batches of a single-component model: calls of at most this many
                           frames run ONE FRAME PER WAVE -- the four rows of a wave share the frame's term loop instead of
                           three of them idling (the reference's one-frame-per-call loop: Shadow DexPilot 0.20 -> 0.16 ms per
                           retarget()).  Same damping rules and trial points up to the summation order of the Hessian: answers
                           agree with the four-frames-per-wave launch to float32 solve accuracy, not bit for bit.  0 off;
                           -1 measured policy (2 048: above that the chip has fewer wave slots than frames)           */
  int32_t sprint_ladder; /* ... and in that launch shape the four rows of a wave may each try THEIR OWN damping value per pass
                           (lambda x 0.03, 0.3, 3, 30): four trial points are evaluated for the instructions of one, the best
                           acceptable one is kept, a pass whose steps are all rejected assembles no model.  Fewer passes on
                           frames that start at an indefinite model (the clean one-frame-per-call regime: Shadow vector 8.5 -> 5.7
                           passes in the host emulation), but ANOTHER iteration than the four-frames-per-wave launch's: answers agree
                           to 1e-4 rad except where a multi-modal frame settles in a different certified minimum.  1 on, 0 off
                           (the rows are copies of one iteration), -1 measured policy.  Round 6: with the ladder on EVERY
                           step is verified by an evaluation at the new point (no unverified last step); the pass that
                           confirms convergence costs kinematics + value only.                                          */
  int32_t tail_passes;  /* sixteen-lane kernel, plain LARGE batches (>= 16 384 frames of a single-component model): the main launch
                           stops every frame after this many passes; the few per cent still unfinished are listed on the device
                           and handed to a second launch in the one-frame-per-wave shape with the ladder above -- a launch is
                           otherwise bound by the passes of its slowest frames, on a chip that is idle by then.  Deterministic
                           per frame (the cap is fixed), but the handed-over frames follow the ladder's iteration from where
                           they stood.  0 off, > 0 the cap, -1 measured policy = OFF: at 65 536 frames the capped main launch
                           is throughput-bound and barely shorter, the second launch comes on top (LEAP position 1.11 ->
                           1.18-1.29 ms for caps of 16 ... 6; answers equal to 1e-6 rad either way)                      */
} dexr_tuning;
#define DEXR_TUNE_LAM_JUMP 1u
#define DEXR_TUNE_LAM_FASTDEC 2u

const char* dexr_last_error(void);
const char* dexr_version(void);
int dexr_device_count(void);
void dexr_default_options(dexr_solve_options* opt);

/* Build a model from a table blob (dexr_model_header + n_comp * dexr_comp_table, see dexr_tables.h).
 * Replaces what Optimizer.__init__ / set_joint_limit / set_kinematic_adaptor cache on the Python side
 * (optimizer.py:18-75) and pin.buildModelFromUrdf (robot_wrapper.py:15). */
int dexr_model_create(const void* blob, size_t nbytes, dexr_model** out);
void dexr_model_destroy(dexr_model* m);
int dexr_model_info(const dexr_model* m, dexr_model_header* header_out);
int dexr_model_get_tuning(const dexr_model* m, dexr_tuning* out);     /* out->struct_size must be set by the caller */
int dexr_model_set_tuning(dexr_model* m, const dexr_tuning* tuning);  /* re-runs the kernel selection           */
/* Which float32 solve kernel the handle launches: DEXR_KERNEL_* in *family, joint bucket in *bucket, 1 in *chain
 * when the serial-chain specialisation is active (2: with its tip pass) (diagnostics for tools/ and tests). */
int dexr_model_kernel(const dexr_model* m, int32_t* family, int32_t* bucket, int32_t* chain);
/* [not-in-ref] Pre-allocate what the `_dev` entry points would otherwise allocate lazily for batches of up to max_batch frames:
 * the hard-frames-first workspaces of the sixteen-lane kernel (dexr_tuning.longest_first; DexPilot batches of >= 32 768 frames
 * by default).  Without it the FIRST such call does a hipMalloc and a call with a larger batch than any before does
 * hipEventSynchronize + hipFree + hipMalloc -- the only place a `_dev` entry point may block on the host; after a reserve()
 * for the largest batch none does.  Under stream capture the ordering is skipped altogether (no allocation, no cross-stream
 * event inside a captured region): a captured graph walks the frames in natural order -- the same answers, bit for bit, on
 * the schedule of longest_first = 0.  No-op for models other kernels serve. */
int dexr_model_reserve(dexr_model* m, int64_t max_batch);
/* Diagnostics: the lane plan of component `comp` for the sixteen-lane kernel -- chain_out[16][16]: lane l, step s ->
 * local joint (bit 7 set when that lane publishes the joint's frame, 0xFF: none); anc_rev_out[DEXR_MAXJ]: revolute
 * ancestors-or-self of each joint.  DEXR_ERR_UNSUPPORTED when the model does not fit that kernel. */
int dexr_model_lane_plan(const dexr_model* m, int32_t comp, int32_t* n_chain, int32_t* depth, uint8_t* chain_out,
                         uint32_t* anc_rev_out);

/* == Optimizer.retarget x B  (optimizer.py:77-102).
 *   ref      B x n_ref x 3 float32  (vector/dexpilot: task-origin vectors; position: target positions)
 *   fixed    B x n_fixed float32    (may be NULL when n_fixed == 0)
 *   last     B x n_opt float32      (start point AND regularisation target, optimizer.py:93,98)
 *   state    B uint32 in/out        (DexPilot projection bits, optimizer.py:466-476; may be NULL)
 *   qpos_out B x n_opt float32      (target_joint_names order)
 *   status_out, iters_out  B int32  (may be NULL)
 *   fval_out B float32              (final f + norm_delta*|x-last|^2, may be NULL)                */
int dexr_retarget_dev(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
                      uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                      const dexr_solve_options* opt, void* stream);
int dexr_retarget(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
                  uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                  const dexr_solve_options* opt);
/* Same solve fed with RAW hand keypoints: keypoints B x n_keypoints x 3 float32 (21 MediaPipe/MANO points); each
 * ref_value row is formed on the fly as kp[task] - kp[origin] (vector, DexPilot) or kp[idx] (position), i.e. the
 * gather/subtract every caller of the reference performs before retarget()
 * (/root/reference/example/profiling/profile_online_retargeting.py:24-30, example/vector_retargeting/detect_from_video.py:45-55). */
int dexr_retarget_kp_dev(const dexr_model* m, int64_t B, const float* keypoints, const float* fixed, const float* last,
                         uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                         const dexr_solve_options* opt, void* stream);
int dexr_retarget_kp(const dexr_model* m, int64_t B, const float* keypoints, const float* fixed, const float* last,
                     uint32_t* state, float* qpos_out, int32_t* status_out, int32_t* iters_out, float* fval_out,
                     const dexr_solve_options* opt);
/* same, float64 arithmetic and float64 result (validation aid; host pointers) */
int dexr_retarget_f64(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
                      uint32_t* state, double* qpos_out, int32_t* status_out, int32_t* iters_out,
                      const dexr_solve_options* opt);

/* == objective(x, grad) x B  (optimizer.py:146-198, 249-304, 510-575): value WITHOUT the norm_delta term,
 * gradient WITH it, float64 arithmetic.  x, grad_out: B x n_opt float64; f_out: B float64.
 * `state` (DexPilot) is read, and updated exactly as get_objective_function's pre-amble does. Host pointers. */
int dexr_eval(const dexr_model* m, int64_t B, const float* ref, const float* fixed, const float* last,
              const double* x, uint32_t* state, double* f_out, double* grad_out);

/* == RobotWrapper.compute_forward_kinematics + get_link_pose(...)[:3,3] x B (robot_wrapper.py:82-87).
 * `m` must come from an FK table (kind DEXR_KIND_FKONLY). q: B x n_q float64 (pinocchio dof order);
 * pos_out: B x n_ref x 3 float64.  Host pointers. */
int dexr_fk(const dexr_model* m, int64_t B, const double* q, double* pos_out);

/* ---- frame sequences: SeqRetargeting.retarget x T frames x B sequences in ONE launch --------
 * Per sequence exactly what /root/reference/src/dex_retargeting/seq_retarget.py:112-124 does per call: clip the carried
 * last_qpos to the joint limits, solve from it (start point and regularisation target), carry the UNFILTERED float32
 * answer -- and the DexPilot projection bits (optimizer.py:466-476) -- to the next frame.  The lane (or quad) that owns
 * a sequence loops over its T frames inside the kernel; nothing visits the host between frames.
 *   inputs       T x B x n_keypoints x 3 float32 raw keypoints (inputs_are_keypoints = 1) or T x B x n_ref x 3 ref_value rows
 *   fixed        T x B x n_fixed float32, or NULL when n_fixed == 0
 *   last_inout   B x n_opt float32: SeqRetargeting.last_qpos of every sequence before frame 0 / after frame T-1
 *   state_inout  B uint32 DexPilot bits (may be NULL)
 *   qpos_raw_out T x B x n_opt float32: the optimiser's answer for every frame (what the reference stores in last_qpos)
 *   status_out   T x B int32 or NULL
 *   joint_limit_eps: the epsilon set_joint_limit widened the box by (optimizer.py:54-60; 1e-3 in the reference): the
 *                carried value is clipped to [lo + eps, hi - eps] of the model's box, i.e. to the joint limits.
 * Models whose default options include the float64 polish launch run their sequences in float64 arithmetic instead
 * (a polish pass cannot be interleaved with the carry).  DEVICE pointers; enqueues on `stream`, no synchronisation. */
int dexr_retarget_seq_dev(const dexr_model* m, int64_t B, int32_t T, const float* inputs, int32_t inputs_are_keypoints,
                          const float* fixed, float* last_inout, uint32_t* state_inout, float* qpos_raw_out,
                          int32_t* status_out, float joint_limit_eps, const dexr_solve_options* opt, void* stream);

/* The rest of SeqRetargeting.retarget for T x B frames (seq_retarget.py:125-133): robot_qpos = zeros; [fixed joints] =
 * fixed_qpos; [target joints] = qpos; mimic joints = source * multiplier + offset (kinematics_adaptor.py:102-105,
 * float64); then LPFilter.next (optimizer_utils.py:7-13): first frame passes through, afterwards y += alpha (x - y).
 *   dof_kind / dof_idx / dof_mult / dof_off: HOST arrays of length n_q (<= DEXR_MAX_DOF) describing every robot dof in
 *     pinocchio order: kind 0 = target joint (idx = column of qpos_raw), 1 = fixed joint (idx = column of fixed),
 *     2 = mimic joint (idx = source dof, value = mult * value(source) + off)
 *   qpos_raw T x B x n_opt float32, fixed T x B x n_fixed float32 or NULL                       (DEVICE)
 *   alpha: low-pass coefficient in [0, 1]; anything else = no filter (retargeting_config.py:232-235)
 *   filter_inout B x n_q float64 = LPFilter.y of every sequence (DEVICE; ignored without a filter);
 *   first_frame_initialises: 1 when the filters have not seen a frame yet (LPFilter.is_init == False)
 *   robot_qpos_out T x B x n_q float64 (DEVICE), the value SeqRetargeting.retarget returns for every frame. */
#define DEXR_MAX_DOF 64
int dexr_seq_compose_dev(int64_t B, int32_t T, int32_t n_q, int32_t n_opt, int32_t n_fixed, const int32_t* dof_kind,
                         const int32_t* dof_idx, const double* dof_mult, const double* dof_off, const float* qpos_raw,
                         const float* fixed, double alpha, double* filter_inout, int32_t first_frame_initialises,
                         double* robot_qpos_out, void* stream);

/* ---- mixed-fleet batches (BASELINE.json configs[4]; SURVEY.md section 8b `dexr_retarget_multi`) -------------------------
 * B frames; frame b is retargeted to models[model_id[b]].  Frames are bucketed by model ON THE DEVICE (count, offsets,
 * index lists -- wavefronts must be model-uniform because the kinematic tables are scalar operands), every model's
 * solve kernel is enqueued over its index list, reading its bucket's size from device memory and reading / writing
 * the caller's rows in place: no gather / scatter copies and no host synchronisation anywhere.
 *   models     n_models <= DEXR_FLEET_MAX_MODELS handles, each with target_link_human_indices (keypoint input)
 *   fixed      B x ld_fixed float32 rows of caller-supplied fixed-joint values (fixed_qpos, optimizer.py:141-142); a model
 *              reads its first n_fixed columns; NULL when no model has fixed joints
 *   model_id   B int32 in [0, n_models); frames with another id are left untouched
 *   keypoints  B x 21 x 3 float32: every model forms its own ref_value rows
 *   last, qpos_out  B x ld float32 rows, ld >= max n_opt; a model reads / writes its first n_opt columns
 *   state      B uint32 in/out (read and written for frames of DexPilot models only; may be NULL)
 *   status_out B int32 or NULL
 *   workspace  dexr_fleet_workspace_bytes(B) bytes of device memory (contents irrelevant).  ALWAYS size it with that function:
 *              since round 4 it also holds the ordering grids of the hard-frames-first walk (about 4x the round-3 formula);
 *              a buffer sized by an older formula is refused with DEXR_ERR_INVALID, never overrun
 * All pointers except `models` are DEVICE pointers; enqueues on `stream`. */
#define DEXR_FLEET_MAX_MODELS 16
size_t dexr_fleet_workspace_bytes(int64_t B);
int dexr_retarget_multi_dev(const dexr_model* const* models, int32_t n_models, int64_t B, const int32_t* model_id,
                            const float* keypoints, const float* fixed, int32_t ld_fixed, const float* last, int32_t ld,
                            uint32_t* state, float* qpos_out, int32_t* status_out, const dexr_solve_options* opt,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The same with HOST pointers (SURVEY.md section 8b `dexr_retarget_multi`): packs, copies, runs, synchronises like the other
 * host entry points (staged through models[0]'s context; the workspace is internal).  qpos_out is in-out: rows of frames
 * with an unknown model id, and the columns beyond a model's n_opt, come back as they were passed. */
int dexr_retarget_multi(const dexr_model* const* models, int32_t n_models, int64_t B, const int32_t* model_id,
                        const float* keypoints, const float* fixed, int32_t ld_fixed, const float* last, int32_t ld,
                        uint32_t* state, float* qpos_out, int32_t* status_out, const dexr_solve_options* opt);

/* ---- multi-GPU: reassembling the qpos tensor (BASELINE.json north_star; SURVEY.md section 8b `dexr_allgather`, 8e) -------
 * One process per GPU.  Frames are independent, so ranks solve contiguous shards with no exchange; the only collective of
 * the path is ONE all-gather of the (B/N, n_opt) result rows.  It runs on RCCL (bound at run time with dlopen: a copy
 * already mapped into the process -- torch's -- is reused, else librccl.so.1 from the loader path / /opt/rocm/lib) and is
 * enqueued on the caller's stream like every "_dev" entry point: solve -> all-gather needs no host round trip and can be
 * captured into one hipGraph.  The reference is single-process (no counterpart).
 *   dexr_comm_unique_id  rank 0 fills DEXR_UNIQUE_ID_BYTES bytes; the HOST application hands them to every rank (any
 *                        transport: a torch.distributed store, MPI, a file)
 *   dexr_comm_create     collective over all ranks; binds the communicator to the CURRENT HIP device
 *   dexr_allgather       recv[r * bytes_per_rank ...] = rank r's `send` block, DEVICE pointers, recv holds world blocks;
 *                        in-place (send == recv + rank * bytes_per_rank) is allowed
 *   dexr_comm_max_f64    control plane: element-wise MAX of n <= 8 HOST doubles over the ranks (timing brackets);
 *                        synchronises `stream`;  dexr_comm_barrier = the same with a dummy value                        */
typedef struct dexr_comm dexr_comm;
#define DEXR_UNIQUE_ID_BYTES 128
int dexr_comm_unique_id(void* id_out);
int dexr_comm_create(const void* unique_id, int32_t rank, int32_t world, dexr_comm** out);
void dexr_comm_destroy(dexr_comm* c);
int dexr_comm_info(const dexr_comm* c, int32_t* rank, int32_t* world, int32_t* rccl_version);
int dexr_allgather(dexr_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int dexr_comm_max_f64(dexr_comm* c, double* values_inout, int32_t n, void* stream);
int dexr_comm_barrier(dexr_comm* c, void* stream);

/* The step right before the path: raw detector keypoints -> wrist-centred keypoints in the MANO frame, x B.
 *   kp_c = kp - kp[0];  R = estimate_frame_from_hand_points(kp_c);  joint_pos = kp_c @ R @ operator2mano
 * (example/vector_retargeting/single_hand_detector.py:102-104,129-158; OPERATOR2MANO_RIGHT/LEFT constants.py:7-21).
 * keypoints, joint_pos_out: B x 21 x 3 float32 (joint_pos_out may alias nothing else); operator2mano: 9 floats,
 * row major, HOST pointer in both variants; wrist_rot_out: B x 3 x 3 float32 (the reference's
 * `mediapipe_wrist_rot`) or NULL.  Frames whose keypoints 0, 5, 9 are collinear have no palm plane: their rows
 * come back non-finite (the solver then reports status 2 for them and keeps last_qpos). */
int dexr_mano_keypoints_dev(int64_t B, const float* keypoints, const float* operator2mano, float* joint_pos_out,
                            float* wrist_rot_out, void* stream);
int dexr_mano_keypoints(int64_t B, const float* keypoints, const float* operator2mano, float* joint_pos_out,
                        float* wrist_rot_out);

#ifdef __cplusplus
}
#endif
#endif /* DEXR_H */
