/* dexr_tables.h -- the compiled kinematic-table format shared by the host (Python table compiler,
 * C++ loader) and the HIP kernels.  Plain C, fixed-size PODs, 4-byte fields only, no padding.
 *
 * One `dexr_model_header` followed by `n_comp` `dexr_comp_table` records is the "model blob" handed to
 * dexr_model_create().  A model is one retargeting problem (robot URDF + optimizer type + link/joint
 * selection + constants), i.e. what RetargetingConfig.build() assembles in the reference
 * (/root/reference/src/dex_retargeting/retargeting_config.py:167-257).  A component is a set of optimised
 * joints that no residual term couples to any other set (e.g. one Allegro finger under VectorOptimizer);
 * components are solved by independent wavefronts.
 *
 * Joint records are in depth-first (pinocchio dof) order restricted to the component.  Every joint frame has
 * been re-aligned by the table compiler so that the joint axis is the local +z axis:
 *   T_k = T_parent(k) * X_k * Rz(q_k)            (revolute)
 *   T_k = T_parent(k) * X_k * Tz(q_k)            (prismatic)
 * with X_k = [R | p] stored row-major as 9 + 3 floats.  T_parent(k) is the running transform (restore = -1),
 * the identity (restore = -2) or a saved slot (restore >= 0); after the joint is processed the transform is
 * saved to slot `save` when save >= 0 (fork points of the tree).
 */
#ifndef DEXR_TABLES_H
#define DEXR_TABLES_H

#include <stdint.h>

#define DEXR_MAGIC 0x52584544u /* "DEXR" */
#define DEXR_TABLE_VERSION 5u

#define DEXR_MAXJ 32  /* joints per component (bitmask width) */
#define DEXR_MAXF 16  /* frames (target links) per component  */
#define DEXR_MAXT 16  /* residual terms per component         */
#define DEXR_NSLOT 3  /* saved transforms (tree fork depth)   */

/* objective kinds (== retargeting_type of the reference's optimizers) */
#define DEXR_KIND_VECTOR 0   /* VectorOptimizer   optimizer.py:203-306 */
#define DEXR_KIND_POSITION 1 /* PositionOptimizer optimizer.py:116-200 */
#define DEXR_KIND_DEXPILOT 2 /* DexPilotOptimizer optimizer.py:309-577 */
#define DEXR_KIND_FKONLY 3   /* forward kinematics of a link list (RobotWrapper.get_link_pose) */

/* where a joint's value comes from */
#define DEXR_SRC_OPT 0    /* optimised variable: x[api]                              */
#define DEXR_SRC_FIXED 1  /* fixed_qpos input:   mult * fixed[src_idx] + off          */
#define DEXR_SRC_MIMIC 2  /* mimic joint:        mult * q[local joint src_idx] + off  */
#define DEXR_SRC_DIRECT 3 /* FK-only tables:     q_full[src_idx]                      */

#define DEXR_JOINT_REVOLUTE 0
#define DEXR_JOINT_PRISMATIC 1

typedef struct dexr_comp_table {
  int32_t n_joint, n_frame, n_term, n_base_frame; /* frames [0,n_base_frame) hang off the fixed base */
  float X[DEXR_MAXJ][12];
  int32_t jtype[DEXR_MAXJ];
  int32_t restore[DEXR_MAXJ];
  int32_t save[DEXR_MAXJ];
  int32_t src_kind[DEXR_MAXJ];
  int32_t src_idx[DEXR_MAXJ];
  int32_t api[DEXR_MAXJ];  /* index into last_qpos / qpos_out rows for DEXR_SRC_OPT joints, else -1 */
  int32_t fbeg[DEXR_MAXJ]; /* frames [fbeg,fend) are rigidly attached to this joint's child body     */
  int32_t fend[DEXR_MAXJ];
  float mult[DEXR_MAXJ];
  float off[DEXR_MAXJ];
  float lo[DEXR_MAXJ]; /* box of the optimised variable (already widened by the reference's 1e-3) */
  float hi[DEXR_MAXJ];
  int32_t frame_joint[DEXR_MAXF]; /* -1 = base */
  float frame_off[DEXR_MAXF][3];  /* position of the frame origin in the (re-aligned) joint frame  */
  uint32_t frame_anc[DEXR_MAXF];  /* bit k set <=> local joint k is an ancestor of the frame       */
  int32_t term_task[DEXR_MAXT];   /* frame index                                                  */
  int32_t term_origin[DEXR_MAXT]; /* frame index, -1 for position terms                           */
  int32_t term_ref[DEXR_MAXT];    /* row of ref_value this term is compared with                  */
  /* Reduced variables (kinematics_adaptor.py:102-113 folded at compile time): the component's optimised variables
   * are numbered 0..n_var-1 in joint order; joint k moves with variable var[k] (its own for an optimised joint, its
   * source's for a mimic joint, -1 for fixed-valued joints) as q_k = vmul[k] * x[var[k]] + off[k], so that
   * dq_k/dx = vmul[k] and the Jacobian column of a variable is the vmul-weighted sum over its joint family.
   * var_joint[v] is the local joint that IS variable v (its api / lo / hi apply). */
  int32_t n_var;
  int32_t var[DEXR_MAXJ];
  float vmul[DEXR_MAXJ];
  int32_t var_joint[DEXR_MAXJ];
} dexr_comp_table;

typedef struct dexr_model_header {
  uint32_t magic, version;
  int32_t kind, n_opt, n_fixed, n_ref, n_comp, num_fingers;
  int32_t n_q;         /* robot dof (FK-only tables: row length of q_full)            */
  int32_t comp_bytes;  /* sizeof(dexr_comp_table), checked by the loader              */
  float huber_delta;   /* SmoothL1 beta                                               */
  float norm_delta;    /* regulariser weight (gradient term 2*norm_delta*(x-last))    */
  float scaling;       /* vector / dexpilot target scaling                            */
  float inv_norm;      /* 1/V (vector, dexpilot) or 1/(3P) (position): 'mean' factor  */
  float project_dist, escape_dist, eta1, eta2; /* DexPilot constants optimizer.py:344-347 */
  /* target_link_human_indices (retargeting_config.py:27-29): how a ref_value row is formed from raw hand keypoints,
   * ref[r] = kp[human_task[r]] - kp[human_origin[r]]  (human_origin[r] = -1: ref[r] = kp[human_task[r]]).
   * n_keypoints = 0 when the model carries no such mapping (keypoint entry points then refuse it). */
  int32_t n_keypoints;
  int32_t human_origin[DEXR_MAXT];
  int32_t human_task[DEXR_MAXT];
} dexr_model_header;

/* ---- generic tables: models that outgrow the fixed-size component records above -----------------------------------------
 * (more than DEXR_MAXJ joints in one component, more than DEXR_MAXT reference rows, more than DEXR_MAXF target links,
 * a tree that forks deeper than DEXR_NSLOT, DexPilot with more than 5 fingers: an arm + hand URDF, a 6-finger hand --
 * anything the reference's Optimizer.__init__ accepts, optimizer.py:18-52).  The blob is then
 *     dexr_model_header (n_comp = 0, comp_bytes = 0)  |  dexr_gen_header  |  the arrays below, in this order,
 * and the model is served by the general kernel (csrc/dexr_gen.hpp: one wavefront per frame, every table in memory,
 * rolled loops, float64) instead of the register / LDS-tiled families.  All joints form ONE component.
 *
 *   double  X[n_joint][12]        placement of joint k in its parent joint's frame (R row-major | p), fixed joints folded
 *   double  axis[n_joint][3]      unit joint axis in the joint's own frame (no re-alignment here)
 *   double  jmul[n_joint], joff[n_joint]   q_k = jmul * x[var[k]] + joff  (var[k] >= 0)  or  jmul * fixed[src_idx[k]] + joff
 *                                          (FK-only tables: q_k = q_full[src_idx[k]])
 *   double  lo[n_var], hi[n_var]  box of the optimised variables (already widened by the reference's 1e-3)
 *   double  frame_off[n_frame][3] frame origin in its joint's frame (world coordinates for frames on the fixed base)
 *   uint64  frame_anc[n_frame]    bit k set <=> joint k is an ancestor of the frame
 *   uint64  joint_anc[n_joint]    bit j set <=> joint j is joint k itself or one of its ancestors
 *   int32   jtype[n_joint], parent[n_joint] (-1 = root), depth[n_joint], src_idx[n_joint], var[n_joint] (-1: not optimised)
 *   int32   var_api[n_var]        column of last_qpos / qpos_out rows
 *   int32   fam_off[n_var + 1], fam[n_fam]   joints that move with each variable (kinematics_adaptor.py:102-113 folded)
 *   int32   frame_joint[n_frame]  (-1 = fixed base)
 *   int32   term_task[n_term], term_origin[n_term] (-1: position term), term_ref[n_term],
 *           row_human_origin[n_term] (-1: ref row r = kp[row_human_task[r]]), row_human_task[n_term]  (indexed by REF ROW;
 *           a generic model has exactly one term per reference row)
 * int32 arrays are padded to a multiple of two entries so that everything stays 8-byte aligned. */
#define DEXR_GEN_MAGIC 0x47584544u /* "DEXG" */
#define DEXR_GEN_MAXJ 64 /* joints, variables, frames and terms of a generic model (one lane each; 64-bit ancestor masks) */
typedef struct dexr_gen_header {
  uint32_t magic;
  int32_t n_joint, n_frame, n_term, n_var, n_fam, max_depth;
  int32_t has_keypoint_map; /* term_human_* filled: raw keypoint input is available */
} dexr_gen_header;

#endif /* DEXR_TABLES_H */
