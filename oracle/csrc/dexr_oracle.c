/* ORACLE (test infrastructure only -- never linked into or called by the product path; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library).
 *
 * Plain-C, float64, one-frame-at-a-time restatement of what ONE call of the reference's objective closure computes
 * (/root/reference/src/dex_retargeting/optimizer.py):
 *   PositionOptimizer  objective(x, grad)   optimizer.py:146-198
 *   VectorOptimizer    objective(x, grad)   optimizer.py:249-304
 *   DexPilotOptimizer  objective(x, grad)   optimizer.py:510-575  (the stateful pre-amble :462-508 stays in
 *                                           oracle/objectives.py: it runs once per frame, not per evaluation)
 * i.e. per evaluation: qpos assembly (fixed joints, target joints, mimic forward -- kinematics_adaptor.py:102-105),
 * forward kinematics of every computed link (robot_wrapper.py:82-87), the world-aligned point Jacobians
 * R_link @ J_local[:3] (optimizer.py:279-284, robot_wrapper.py:93-95), SmoothL1 (of the vector norm for vector /
 * DexPilot, per coordinate for position), the chain rule through the Jacobians, the mimic fold
 * (kinematics_adaptor.py:107-113) and the 2 norm_delta (x - last) term that the GRADIENT carries while the returned
 * VALUE does not (optimizer.py:194-198, 300-304, 571-575).
 *
 * It restates the same mathematics as oracle/objectives.py + oracle/kin.py (numpy, batched) and is pinned to the
 * same vectors: tests/golden/objective_golden.npz, produced by the reference's OWN closures (tests/test_oracle_c.py).
 * Its purpose is the CPU baseline: a compiled per-evaluation cost comparable to pinocchio's, driven by scipy's
 * compiled SLSQP (nlopt is not installed), instead of numpy's per-call interpreter overhead.
 *
 * Kinematics convention (as oracle/kin.py): a link's world pose is the product along its chain of
 * origin(R0, p0) * motion(axis, q); the chain is walked from the root for every computed link.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXCH 64 /* movable + fixed joints on one root-to-link chain */

typedef struct {
  int kind; /* 0 vector, 1 position, 2 dexpilot */
  int n_q, n_opt, n_fixed, n_links, n_ref, n_mimic, n_chain_joints;
  double huber_delta, norm_delta;
  int *idx_pin2target, *idx_pin2fixed;
  int *mimic_pin, *mimic_src_pin, *mimic_target2source;
  double *mimic_mult, *mimic_off;
  int *chain_off;  /* n_links + 1 */
  int *jtype;      /* per chain joint: 0 fixed, 1 revolute, 2 prismatic */
  int *jq;         /* per chain joint: index into q (-1 for fixed) */
  double *R0, *p0, *axis; /* per chain joint: 9, 3, 3 */
  int *origin_idx, *task_idx; /* n_ref (vector / dexpilot) */
  double *q, *pos, *jac, *gpos; /* work: n_q, n_links*3, n_links*3*n_q, n_links*3 */
} oracle_model;

static void *dup_mem(const void *src, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (p && bytes) memcpy(p, src, bytes);
  return p;
}

oracle_model *oracle_create(int kind, int n_q, int n_opt, int n_fixed, int n_links, int n_ref, int n_mimic,
                            double huber_delta, double norm_delta, const int *idx_pin2target, const int *idx_pin2fixed,
                            const int *mimic_pin, const int *mimic_src_pin, const int *mimic_target2source,
                            const double *mimic_mult, const double *mimic_off, const int *chain_off, const int *jtype,
                            const int *jq, const double *R0, const double *p0, const double *axis,
                            const int *origin_idx, const int *task_idx) {
  oracle_model *m = (oracle_model *)calloc(1, sizeof(oracle_model));
  if (!m) return NULL;
  const int nc = chain_off[n_links];
  if (n_q > 512) {
    free(m);
    return NULL;
  }
  for (int l = 0; l < n_links; ++l)
    if (chain_off[l + 1] - chain_off[l] > MAXCH) {
      free(m);
      return NULL;
    }
  m->kind = kind; m->n_q = n_q; m->n_opt = n_opt; m->n_fixed = n_fixed; m->n_links = n_links; m->n_ref = n_ref;
  m->n_mimic = n_mimic; m->n_chain_joints = nc; m->huber_delta = huber_delta; m->norm_delta = norm_delta;
  m->idx_pin2target = (int *)dup_mem(idx_pin2target, sizeof(int) * n_opt);
  m->idx_pin2fixed = (int *)dup_mem(idx_pin2fixed, sizeof(int) * n_fixed);
  m->mimic_pin = (int *)dup_mem(mimic_pin, sizeof(int) * n_mimic);
  m->mimic_src_pin = (int *)dup_mem(mimic_src_pin, sizeof(int) * n_mimic);
  m->mimic_target2source = (int *)dup_mem(mimic_target2source, sizeof(int) * n_mimic);
  m->mimic_mult = (double *)dup_mem(mimic_mult, sizeof(double) * n_mimic);
  m->mimic_off = (double *)dup_mem(mimic_off, sizeof(double) * n_mimic);
  m->chain_off = (int *)dup_mem(chain_off, sizeof(int) * (n_links + 1));
  m->jtype = (int *)dup_mem(jtype, sizeof(int) * nc);
  m->jq = (int *)dup_mem(jq, sizeof(int) * nc);
  m->R0 = (double *)dup_mem(R0, sizeof(double) * 9 * nc);
  m->p0 = (double *)dup_mem(p0, sizeof(double) * 3 * nc);
  m->axis = (double *)dup_mem(axis, sizeof(double) * 3 * nc);
  m->origin_idx = (int *)dup_mem(origin_idx, sizeof(int) * (kind == 1 ? 0 : n_ref));
  m->task_idx = (int *)dup_mem(task_idx, sizeof(int) * (kind == 1 ? 0 : n_ref));
  m->q = (double *)calloc((size_t)n_q + 1, sizeof(double));
  m->pos = (double *)calloc((size_t)n_links * 3 + 1, sizeof(double));
  m->jac = (double *)calloc((size_t)n_links * 3 * n_q + 1, sizeof(double));
  m->gpos = (double *)calloc((size_t)n_links * 3 + 1, sizeof(double));
  return m;
}

void oracle_destroy(oracle_model *m) {
  if (!m) return;
  free(m->idx_pin2target); free(m->idx_pin2fixed); free(m->mimic_pin); free(m->mimic_src_pin);
  free(m->mimic_target2source); free(m->mimic_mult); free(m->mimic_off); free(m->chain_off); free(m->jtype);
  free(m->jq); free(m->R0); free(m->p0); free(m->axis); free(m->origin_idx); free(m->task_idx);
  free(m->q); free(m->pos); free(m->jac); free(m->gpos);
  free(m);
}

static void mat3_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* forward kinematics + world-aligned point Jacobian of one link (oracle/kin.py: _walk, point_jacobians) */
static void link_fk(const oracle_model *m, int l, const double *q, double *p_out, double *jac_out /* 3 x n_q or NULL */) {
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0}, T[9];
  double aw[MAXCH][3], ow[MAXCH][3];
  int qi[MAXCH], ty[MAXCH], n = 0;
  for (int c = m->chain_off[l]; c < m->chain_off[l + 1]; ++c) {
    const double *R0 = m->R0 + 9 * c, *p0 = m->p0 + 3 * c, *ax = m->axis + 3 * c;
    for (int i = 0; i < 3; ++i) p[i] += R[3 * i] * p0[0] + R[3 * i + 1] * p0[1] + R[3 * i + 2] * p0[2];
    mat3_mul(R, R0, T);
    memcpy(R, T, sizeof(R));
    if (m->jtype[c] == 0) continue;
    for (int i = 0; i < 3; ++i) {
      aw[n][i] = R[3 * i] * ax[0] + R[3 * i + 1] * ax[1] + R[3 * i + 2] * ax[2];
      ow[n][i] = p[i];
    }
    qi[n] = m->jq[c];
    ty[n] = m->jtype[c];
    const double th = q[m->jq[c]];
    if (m->jtype[c] == 1) { /* Rodrigues: I + sin K + (1 - cos) K^2 */
      const double s = sin(th), k = 1.0 - cos(th);
      const double K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
      double K2[9], Rq[9];
      mat3_mul(K, K, K2);
      for (int i = 0; i < 9; ++i) Rq[i] = s * K[i] + k * K2[i];
      Rq[0] += 1.0; Rq[4] += 1.0; Rq[8] += 1.0;
      mat3_mul(R, Rq, T);
      memcpy(R, T, sizeof(R));
    } else {
      for (int i = 0; i < 3; ++i) p[i] += aw[n][i] * th;
    }
    ++n;
  }
  p_out[0] = p[0]; p_out[1] = p[1]; p_out[2] = p[2];
  if (!jac_out) return;
  memset(jac_out, 0, sizeof(double) * 3 * m->n_q);
  for (int k = 0; k < n; ++k) {
    double col[3];
    if (ty[k] == 1) {
      const double d[3] = {p[0] - ow[k][0], p[1] - ow[k][1], p[2] - ow[k][2]};
      cross3(aw[k], d, col);
    } else {
      col[0] = aw[k][0]; col[1] = aw[k][1]; col[2] = aw[k][2];
    }
    for (int i = 0; i < 3; ++i) jac_out[i * m->n_q + qi[k]] = col[i];
  }
}

/* torch.nn.SmoothL1Loss(beta) element value / derivative for input d vs target 0 */
static double smooth_l1(double d, double beta, double *der) {
  const double ad = fabs(d);
  if (ad < beta) {
    *der = d / beta;
    return 0.5 * d * d / beta;
  }
  *der = d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0);
  return ad - 0.5 * beta;
}

/* One evaluation of the closure.  x: n_opt; tgt: n_ref x 3 float64 (vector: ref_value * scaling as the caller formed it;
 * dexpilot: the pre-amble's reference vectors; position: target positions); fixed: n_fixed or NULL; last: n_opt or NULL
 * (no regulariser term in the gradient); weights: n_ref (dexpilot) or NULL (ones); grad: n_opt or NULL.  Returns the
 * value WITHOUT the norm_delta term. */
double oracle_evaluate(oracle_model *m, const double *x, const double *tgt, const double *fixed, const double *last,
                       const double *weights, double *grad) {
  double *q = m->q;
  memset(q, 0, sizeof(double) * m->n_q);
  for (int i = 0; i < m->n_fixed; ++i) q[m->idx_pin2fixed[i]] = fixed[i];
  for (int i = 0; i < m->n_opt; ++i) q[m->idx_pin2target[i]] = x[i];
  for (int i = 0; i < m->n_mimic; ++i) q[m->mimic_pin[i]] = q[m->mimic_src_pin[i]] * m->mimic_mult[i] + m->mimic_off[i];
  for (int l = 0; l < m->n_links; ++l)
    link_fk(m, l, q, m->pos + 3 * l, grad ? m->jac + (size_t)l * 3 * m->n_q : NULL);
  memset(m->gpos, 0, sizeof(double) * 3 * m->n_links);
  const double beta = m->huber_delta;
  double f = 0.0;
  if (m->kind == 1) {
    const int n = m->n_links * 3;
    for (int i = 0; i < n; ++i) {
      double der;
      f += smooth_l1(m->pos[i] - tgt[i], beta, &der);
      m->gpos[i] = der / n;
    }
    f /= n;
  } else {
    const int V = m->n_ref;
    for (int t = 0; t < V; ++t) {
      const double *pt = m->pos + 3 * m->task_idx[t], *po = m->pos + 3 * m->origin_idx[t];
      const double diff[3] = {pt[0] - po[0] - tgt[3 * t], pt[1] - po[1] - tgt[3 * t + 1], pt[2] - po[2] - tgt[3 * t + 2]};
      const double d = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
      const double w = weights ? weights[t] : 1.0;
      double der;
      f += smooth_l1(d, beta, &der) * w;
      if (d > 0) { /* torch.norm backward: zero sub-gradient at 0 */
        const double s = der * w / V / d;
        for (int i = 0; i < 3; ++i) {
          m->gpos[3 * m->task_idx[t] + i] += s * diff[i];
          m->gpos[3 * m->origin_idx[t] + i] -= s * diff[i];
        }
      }
    }
    f /= V;
  }
  if (!grad) return f;
  /* chain rule through the point Jacobians, over ALL dofs; then select / fold (optimizer.py:293-296) */
  double gq[512];
  const int nq = m->n_q;
  for (int j = 0; j < nq; ++j) gq[j] = 0.0;
  for (int l = 0; l < m->n_links; ++l)
    for (int i = 0; i < 3; ++i) {
      const double g = m->gpos[3 * l + i];
      if (g == 0.0) continue;
      const double *row = m->jac + ((size_t)l * 3 + i) * nq;
      for (int j = 0; j < nq; ++j) gq[j] += g * row[j];
    }
  for (int i = 0; i < m->n_opt; ++i) grad[i] = gq[m->idx_pin2target[i]];
  for (int i = 0; i < m->n_mimic; ++i) grad[m->mimic_target2source[i]] += gq[m->mimic_pin[i]] * m->mimic_mult[i];
  if (last)
    for (int i = 0; i < m->n_opt; ++i) grad[i] += 2.0 * m->norm_delta * (x[i] - last[i]);
  return f;
}

/* link positions only (pins the kinematics against oracle/kin.py and fk_golden.npz): q over all n_q dofs */
void oracle_link_positions(oracle_model *m, const double *q, double *pos_out) {
  for (int l = 0; l < m->n_links; ++l) link_fk(m, l, q, pos_out + 3 * l, NULL);
}

int oracle_max_dofs(void) { return 512; }
