/* mzo_physics.c — CPU ORACLE (test infrastructure, not the product).
 *
 * A from-scratch float64 restatement of the MuJoCo subset that the reference's
 * hot path executes inside `mj_step` (called from mujoco_maze/ant.py:63 via
 * gym's do_simulation, mujoco_maze/point.py:57-59, swimmer.py:39):
 * kinematics, joint-space inertia (composite rigid body), bias forces
 * (recursive Newton-Euler), passive damping, motor actuation with ctrl clamp,
 * collision (plane-sphere, plane-capsule, sphere-box, capsule-box, capsule-capsule), joint-limit
 * and pyramidal-cone contact constraints with MuJoCo's impedance / reference-
 * acceleration model, the primal Newton solver, and RK4 integration with
 * manifold quaternion update.
 *
 * PARITY UNPINNED for this file: MuJoCo itself (mujoco-py 2.0.2.13 / MuJoCo 2.0
 * per the reference's poetry.lock:145-146; `mujoco` >= 2.2 per point.py:12) is a
 * third-party dependency that is absent from /root/reference and from this
 * image, and the reference's tests pin no numeric physics value
 * (tests/test_envs.py asserts shapes and reward sign only).  The algorithm below
 * restates MuJoCo's published computation model ("Computation" chapter:
 * constraint model, impedance, reference acceleration, pyramidal cones, Newton
 * solver, RK4); every place where a detail is an assumption is tagged
 * [ASSUME-n] and listed in DESIGN.md.  It is validated by physical invariants
 * (tests/test_oracle_physics.py) and cross-checked against an independent numpy
 * mass matrix (mujoco_maze_amd/model.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mzo.h"

#define NB MZ_MAX_BODY
#define ND MZ_MAX_DOF
#define MINVAL 1e-15
#define MZO_BOX_MINOVERLAP 1e-6 /* box-box: smallest width of the face intersection that still makes contacts (box_box) */

/* ------------------------------------------------------------------ vec helpers */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void cpy3(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void sub3(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void add3(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void addscl3(double* r, const double* a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static inline void mulmat3vec(double* r, const double* m, const double* v) { /* row-major 3x3 */
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmat3Tvec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_to_mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void axis_angle_quat(double* q, const double* axis, double angle) {
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* ------------------------------------------------------------------ kinematics */
static void kinematics(const mz_model* m, mzo_data* d) {
  memset(d->xpos[0], 0, sizeof(d->xpos[0]));
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat_to_mat(d->xmat[0], d->xquat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double pos[3], quat[4];
    int j0 = m->body_jntadr[b], jn = m->body_jntnum[b];
    if (jn == 1 && m->jnt_type[j0] == MZ_JNT_FREE) {
      int qa = m->jnt_qposadr[j0];
      cpy3(pos, d->qpos + qa);
      memcpy(quat, d->qpos + qa + 3, sizeof(quat));
      quat_normalize(quat);
      cpy3(d->xanchor[j0], pos);
      cpy3(d->xaxis[j0], m->jnt_axis[j0]);
    } else {
      double t[3];
      mulmat3vec(t, d->xmat[p], m->body_pos[b]);
      add3(pos, d->xpos[p], t);
      quat_mul(quat, d->xquat[p], m->body_quat[b]);
      for (int j = j0; j < j0 + jn; j++) {
        double mat[9], axis[3], anchor[3];
        quat_to_mat(mat, quat);
        mulmat3vec(axis, mat, m->jnt_axis[j]);
        mulmat3vec(anchor, mat, m->jnt_pos[j]);
        add3(anchor, anchor, pos);
        double q = d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
        if (m->jnt_type[j] == MZ_JNT_SLIDE) {
          addscl3(pos, axis, q);
        } else if (m->jnt_type[j] == MZ_JNT_BALL) {
          /* ball (mj_kinematics): the joint's coordinates ARE the relative quaternion (normalised in place, as for the free
           * joint [ASSUME-8]); the body turns about the anchor */
          double ql[4], qn[4], v[3];
          memcpy(ql, d->qpos + m->jnt_qposadr[j], sizeof(ql));
          quat_normalize(ql);
          quat_mul(qn, quat, ql);
          memcpy(quat, qn, sizeof(quat));
          quat_to_mat(mat, quat);
          mulmat3vec(v, mat, m->jnt_pos[j]);
          sub3(pos, anchor, v);
        } else { /* hinge: rotate about the joint axis through the anchor */
          double ql[4], qn[4], v[3];
          axis_angle_quat(ql, m->jnt_axis[j], q);
          quat_mul(qn, quat, ql);
          memcpy(quat, qn, sizeof(quat));
          quat_to_mat(mat, quat);
          mulmat3vec(v, mat, m->jnt_pos[j]);
          sub3(pos, anchor, v);
        }
        cpy3(d->xaxis[j], axis);
        cpy3(d->xanchor[j], anchor);
      }
    }
    quat_normalize(quat);
    cpy3(d->xpos[b], pos);
    memcpy(d->xquat[b], quat, sizeof(quat));
    quat_to_mat(d->xmat[b], quat);
    double t[3];
    mulmat3vec(t, d->xmat[b], m->body_ipos[b]);
    add3(d->xipos[b], d->xpos[b], t);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], q[4];
    mulmat3vec(t, d->xmat[b], m->geom_pos[g]);
    add3(d->geom_xpos[g], d->xpos[b], t);
    quat_mul(q, d->xquat[b], m->geom_quat[g]);
    quat_to_mat(d->geom_xmat[g], q);
  }
}

/* ------------------------------------------------------------------ spatial algebra
 * 6-vectors are [angular(3); linear(3)], world orientation, taken at the fixed
 * reference point c = origin of body 1 at this instant. */
static void motion_cross(double* r, const double* v, const double* s) {
  double a[3], b[3], c[3];
  cross3(a, v, s);
  cross3(b, v, s + 3);
  cross3(c, v + 3, s);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void force_cross(double* r, const double* v, const double* f) { /* v x* f */
  double a[3], b[3], c[3];
  cross3(a, v, f);
  cross3(b, v + 3, f + 3);
  cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
/* compact spatial inertia: [mass, h(3) = m*r, Ibar(6) = xx yy zz xy xz yz about c] */
static void inertia_mul(double* r, const double* I, const double* v) {
  const double* h = I + 1;
  const double* J = I + 4;
  double w[3] = {v[0], v[1], v[2]}, l[3] = {v[3], v[4], v[5]}, a[3], b[3];
  cross3(a, h, l);
  cross3(b, w, h);
  r[0] = J[0] * w[0] + J[3] * w[1] + J[4] * w[2] + a[0];
  r[1] = J[3] * w[0] + J[1] * w[1] + J[5] * w[2] + a[1];
  r[2] = J[4] * w[0] + J[5] * w[1] + J[2] * w[2] + a[2];
  r[3] = I[0] * l[0] + b[0]; r[4] = I[0] * l[1] + b[1]; r[5] = I[0] * l[2] + b[2];
}
static inline double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

static int dof_parent(const mz_model* m, int i) { /* previous dof up the kinematic tree, -1 at the root */
  int b = m->dof_bodyid[i];
  if (i > m->body_dofadr[b]) return i - 1;
  b = m->body_parent[b];
  while (b > 0) {
    if (m->body_dofnum[b] > 0) return m->body_dofadr[b] + m->body_dofnum[b] - 1;
    b = m->body_parent[b];
  }
  return -1;
}

/* motion axes S, body spatial inertias, CRBA mass matrix */
static void com_and_crb(const mz_model* m, mzo_data* d) {
  const double* c = d->xpos[1];
  cpy3(d->refpoint, c);
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], d0 = m->jnt_dofadr[j];
    double off[3];
    if (m->jnt_type[j] == MZ_JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        memset(d->S[d0 + k], 0, 6 * sizeof(double));
        d->S[d0 + k][3 + k] = 1.0;
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        sub3(off, c, d->xpos[b]);
        cpy3(d->S[d0 + 3 + k], ax);
        cross3(d->S[d0 + 3 + k] + 3, ax, off);
      }
    } else if (m->jnt_type[j] == MZ_JNT_BALL) {
      /* three rotations about the body's own axes (the body frame AFTER all of its joints) through the anchor: the dofs are the
       * angular velocity in the child frame, as the free joint's rotational half */
      sub3(off, c, d->xanchor[j]);
      for (int k = 0; k < 3; k++) {
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        cpy3(d->S[d0 + k], ax);
        cross3(d->S[d0 + k] + 3, ax, off);
      }
    } else if (m->jnt_type[j] == MZ_JNT_SLIDE) {
      memset(d->S[d0], 0, 3 * sizeof(double));
      cpy3(d->S[d0] + 3, d->xaxis[j]);
    } else {
      sub3(off, c, d->xanchor[j]);
      cpy3(d->S[d0], d->xaxis[j]);
      cross3(d->S[d0] + 3, d->xaxis[j], off);
    }
  }
  for (int b = 1; b < m->nbody; b++) {
    double r[3], Iw[9], tmp[9];
    const double* I6 = m->body_inertia[b];
    double Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    const double* R = d->xmat[b];
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) tmp[3 * i + k] = R[3 * i] * Ib[k] + R[3 * i + 1] * Ib[3 + k] + R[3 * i + 2] * Ib[6 + k];
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) Iw[3 * i + k] = tmp[3 * i] * R[3 * k] + tmp[3 * i + 1] * R[3 * k + 1] + tmp[3 * i + 2] * R[3 * k + 2];
    sub3(r, d->xipos[b], c);
    double ms = m->body_mass[b], rr = dot3(r, r);
    double* I = d->cinert[b];
    I[0] = ms; I[1] = ms * r[0]; I[2] = ms * r[1]; I[3] = ms * r[2];
    I[4] = Iw[0] + ms * (rr - r[0] * r[0]);
    I[5] = Iw[4] + ms * (rr - r[1] * r[1]);
    I[6] = Iw[8] + ms * (rr - r[2] * r[2]);
    I[7] = Iw[1] - ms * r[0] * r[1];
    I[8] = Iw[2] - ms * r[0] * r[2];
    I[9] = Iw[5] - ms * r[1] * r[2];
  }
  /* composite inertias */
  double crb[NB][10];
  memcpy(crb, d->cinert, sizeof(crb));
  for (int b = m->nbody - 1; b >= 1; b--) {
    int p = m->body_parent[b];
    if (p > 0)
      for (int k = 0; k < 10; k++) crb[p][k] += crb[b][k];
  }
  int nv = m->nv;
  for (int i = 0; i < nv; i++)
    for (int j = 0; j < nv; j++) d->M[i][j] = 0.0;
  for (int i = 0; i < nv; i++) {
    double F[6];
    inertia_mul(F, crb[m->dof_bodyid[i]], d->S[i]);
    d->M[i][i] = dot6(d->S[i], F) + m->dof_armature[i];
    for (int j = dof_parent(m, i); j >= 0; j = dof_parent(m, j)) {
      double v = dot6(d->S[j], F);
      d->M[i][j] = v; d->M[j][i] = v;
    }
  }
}

/* dense Cholesky A = L L^T (lower), returns 0 on success */
static int chol_factor(double A[][ND], int n) {
  for (int j = 0; j < n; j++) {
    double s = A[j][j];
    for (int k = 0; k < j; k++) s -= A[j][k] * A[j][k];
    if (s < MINVAL) return -1;
    A[j][j] = sqrt(s);
    for (int i = j + 1; i < n; i++) {
      double t = A[i][j];
      for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k];
      A[i][j] = t / A[j][j];
    }
  }
  return 0;
}
static void chol_solve(double A[][ND], int n, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= A[i][k] * x[k];
    x[i] = s / A[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= A[k][i] * x[k];
    x[i] = s / A[i][i];
  }
}

/* velocities, bias forces via RNE with qacc = 0, passive, actuation (M7) */
static void velocity_and_forces(const mz_model* m, mzo_data* d, const double* ctrl) {
  double cvel[NB][6], cacc[NB][6], cfrc[NB][6];
  memset(cvel[0], 0, sizeof(cvel[0]));
  memset(cacc[0], 0, sizeof(cacc[0]));
  cacc[0][3] = -m->gravity[0]; cacc[0][4] = -m->gravity[1]; cacc[0][5] = -m->gravity[2];
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double v[6], a[6];
    memcpy(v, cvel[p], sizeof(v));
    memcpy(a, cacc[p], sizeof(a));
    int j0 = m->body_jntadr[b];
    for (int j = j0; j < j0 + m->body_jntnum[b]; j++) {
      int d0 = m->jnt_dofadr[j];
      if (m->jnt_type[j] == MZ_JNT_FREE) {
        /* translational axes are world-fixed: Sdot = 0 */
        for (int k = 0; k < 3; k++)
          for (int e = 0; e < 6; e++) v[e] += d->S[d0 + k][e] * d->qvel[d0 + k];
        double sd[3][6];
        for (int k = 0; k < 3; k++) motion_cross(sd[k], v, d->S[d0 + 3 + k]);
        for (int k = 0; k < 3; k++)
          for (int e = 0; e < 6; e++) {
            a[e] += sd[k][e] * d->qvel[d0 + 3 + k];
            v[e] += d->S[d0 + 3 + k][e] * d->qvel[d0 + 3 + k];
          }
      } else if (m->jnt_type[j] == MZ_JNT_BALL) {
        /* mj_comVel: all three axis derivatives from the velocity in front of the joint, then the joint's velocity */
        double sd[3][6];
        for (int k = 0; k < 3; k++) motion_cross(sd[k], v, d->S[d0 + k]);
        for (int k = 0; k < 3; k++)
          for (int e = 0; e < 6; e++) {
            a[e] += sd[k][e] * d->qvel[d0 + k];
            v[e] += d->S[d0 + k][e] * d->qvel[d0 + k];
          }
      } else {
        double sd[6];
        motion_cross(sd, v, d->S[d0]);
        for (int e = 0; e < 6; e++) {
          a[e] += sd[e] * d->qvel[d0];
          v[e] += d->S[d0][e] * d->qvel[d0];
        }
      }
    }
    memcpy(cvel[b], v, sizeof(v));
    memcpy(cacc[b], a, sizeof(a));
    double Ia[6], Iv[6], vf[6];
    inertia_mul(Ia, d->cinert[b], a);
    inertia_mul(Iv, d->cinert[b], v);
    force_cross(vf, v, Iv);
    for (int e = 0; e < 6; e++) cfrc[b][e] = Ia[e] + vf[e];
  }
  memcpy(d->cvel, cvel, sizeof(cvel));
  for (int b = m->nbody - 1; b >= 1; b--) {
    int p = m->body_parent[b];
    if (p > 0)
      for (int e = 0; e < 6; e++) cfrc[p][e] += cfrc[b][e];
  }
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_bias[i] = dot6(d->S[i], cfrc[m->dof_bodyid[i]]);
    d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
    d->qfrc_actuator[i] = 0.0;
  }
  /* joint springs (mj_passive: -stiffness * (qpos - qpos_spring)), hinge and slide joints */
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_stiffness[j] != 0.0 && (m->jnt_type[j] == MZ_JNT_HINGE || m->jnt_type[j] == MZ_JNT_SLIDE))
      d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]] - m->jnt_springref[j]);
  /* inertia-box fluid model (swimmer): [ASSUME-9] only when density/viscosity > 0 */
  if (m->density > 0.0 || m->viscosity > 0.0) mzo_fluid_passive(m, d);
  for (int a = 0; a < m->nu; a++) {
    double u = ctrl ? ctrl[a] : 0.0;
    if (m->act_ctrllimited[a]) {
      if (u < m->act_ctrlrange[a][0]) u = m->act_ctrlrange[a][0];
      if (u > m->act_ctrlrange[a][1]) u = m->act_ctrlrange[a][1];
    }
    /* mj_fwdActuation: force = gain * ctrl + bias0 + bias1 * length + bias2 * velocity (length = gear * q of the joint); motors: gain 1, bias 0 */
    const int dof = m->act_dofid[a], jq = m->jnt_qposadr[m->dof_jntid[dof]];
    const double len = m->act_gear[a] * (d->qpos[jq] - m->qpos0[jq]), vel = m->act_gear[a] * d->qvel[dof];
    const double force = m->act_gainprm[a] * u + m->act_biasprm[a][0] + m->act_biasprm[a][1] * len + m->act_biasprm[a][2] * vel;
    d->qfrc_actuator[dof] += m->act_gear[a] * force;
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
}

/* MuJoCo's "inertia box" fluid force model for bodies moving in a medium
 * (option density / viscosity; swimmer.xml:3).  [ASSUME-9] restated from the
 * published description: each body is replaced by the box with the same mass and
 * principal inertia; viscous drag  f = -3 beta pi d_eq v,  tau = -beta pi d_eq^3 w;
 * quadratic drag per axis  f_i = -rho/2 * A_i |v_i| v_i  etc.  Expressed in the
 * body's inertial frame and mapped back through the COM Jacobian. */
void mzo_fluid_passive(const mz_model* m, mzo_data* d) {
  const double* c = d->refpoint;
  for (int b = 1; b < m->nbody; b++) {
    double mass = m->body_mass[b];
    if (mass < MINVAL) continue;
    /* principal moments in the inertial frame ximat = xmat * R(body_iquat), as MuJoCo (mjModel.body_inertia / body_iquat):
       the diagonal of the body-frame tensor is something else for a tilted link */
    const double* I = m->body_pinertia[b];
    double Ri[9], xim[9];
    quat_to_mat(Ri, m->body_iquat[b]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) xim[3 * i + j] = d->xmat[b][3 * i] * Ri[j] + d->xmat[b][3 * i + 1] * Ri[3 + j] + d->xmat[b][3 * i + 2] * Ri[6 + j];
    double bx[3];
    bx[0] = sqrt(fmax(MINVAL, (I[1] + I[2] - I[0])) / mass * 6.0);
    bx[1] = sqrt(fmax(MINVAL, (I[0] + I[2] - I[1])) / mass * 6.0);
    bx[2] = sqrt(fmax(MINVAL, (I[0] + I[1] - I[2])) / mass * 6.0);
    /* velocity of the COM and angular velocity, local frame */
    double w[3] = {d->cvel[b][0], d->cvel[b][1], d->cvel[b][2]}, r[3], wr[3], vc[3], lw[3], lv[3];
    sub3(r, d->xipos[b], c);
    cross3(wr, w, r);
    vc[0] = d->cvel[b][3] + wr[0]; vc[1] = d->cvel[b][4] + wr[1]; vc[2] = d->cvel[b][5] + wr[2];
    mulmat3Tvec(lw, xim, w);
    mulmat3Tvec(lv, xim, vc);
    double lf[3] = {0, 0, 0}, lt[3] = {0, 0, 0};
    if (m->viscosity > 0.0) {
      double diam = (bx[0] + bx[1] + bx[2]) / 3.0;
      double s = -3.0 * m->viscosity * M_PI * diam;
      double t = -m->viscosity * M_PI * diam * diam * diam;
      for (int k = 0; k < 3; k++) { lf[k] += s * lv[k]; lt[k] += t * lw[k]; }
    }
    if (m->density > 0.0) {
      lf[0] -= 0.5 * m->density * bx[1] * bx[2] * fabs(lv[0]) * lv[0];
      lf[1] -= 0.5 * m->density * bx[0] * bx[2] * fabs(lv[1]) * lv[1];
      lf[2] -= 0.5 * m->density * bx[0] * bx[1] * fabs(lv[2]) * lv[2];
      lt[0] -= m->density * bx[0] * (pow(bx[1], 4) + pow(bx[2], 4)) * fabs(lw[0]) * lw[0] / 64.0;
      lt[1] -= m->density * bx[1] * (pow(bx[0], 4) + pow(bx[2], 4)) * fabs(lw[1]) * lw[1] / 64.0;
      lt[2] -= m->density * bx[2] * (pow(bx[0], 4) + pow(bx[1], 4)) * fabs(lw[2]) * lw[2] / 64.0;
    }
    double f[3], t[3], fsp[6], rf[3];
    mulmat3vec(f, xim, lf);
    mulmat3vec(t, xim, lt);
    /* wrench at c: torque about c = t + r x f */
    cross3(rf, r, f);
    fsp[0] = t[0] + rf[0]; fsp[1] = t[1] + rf[1]; fsp[2] = t[2] + rf[2];
    fsp[3] = f[0]; fsp[4] = f[1]; fsp[5] = f[2];
    for (int i = 0; i < m->nv; i++) {
      /* dof i moves body b iff body(i) is b or an ancestor */
      int a = b, hit = 0;
      while (a > 0) { if (a == m->dof_bodyid[i]) { hit = 1; break; } a = m->body_parent[a]; }
      if (hit) d->qfrc_passive[i] += dot6(d->S[i], fsp);
    }
  }
}

/* ------------------------------------------------------------------ collision */
static void make_frame(double* fr) { /* fr[0:3] normal given, fr[3:6] optional hint */
  double n = norm3(fr);
  fr[0] /= n; fr[1] /= n; fr[2] /= n;
  if (norm3(fr + 3) < 0.5) {
    fr[3] = 0; fr[4] = 0; fr[5] = 0;
    if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1; else fr[5] = 1;
  }
  double t = dot3(fr, fr + 3);
  addscl3(fr + 3, fr, -t);
  n = norm3(fr + 3);
  if (n < 1e-10) { /* hint parallel to the normal: fall back to the default rule */
    fr[3] = 0; fr[4] = 0; fr[5] = 0;
    if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1; else fr[5] = 1;
    t = dot3(fr, fr + 3);
    addscl3(fr + 3, fr, -t);
    n = norm3(fr + 3);
  }
  fr[3] /= n; fr[4] /= n; fr[5] /= n;
  cross3(fr + 6, fr, fr + 3);
}

typedef struct { /* mixed pair parameters */
  double margin, gap, mu, solref[2], solimp[5];
  int condim, b1, b2, g1, g2;
} pairparam;

static void add_contact(mzo_data* d, const pairparam* pp, double dist, const double* pos, const double* normal,
                        const double* hint) {
  if (d->ncon >= MZO_MAX_CON) { d->status |= MZ_STATUS_CONTACT_OVERFLOW; return; }
  mzo_contact* c = &d->con[d->ncon++];
  c->dist = dist;
  cpy3(c->pos, pos);
  cpy3(c->frame, normal);
  if (hint) cpy3(c->frame + 3, hint); else c->frame[3] = c->frame[4] = c->frame[5] = 0;
  make_frame(c->frame);
  c->includemargin = pp->margin - pp->gap;
  c->mu = pp->mu;
  memcpy(c->solref, pp->solref, sizeof(c->solref));
  memcpy(c->solimp, pp->solimp, sizeof(c->solimp));
  c->dim = pp->condim;
  c->body1 = pp->b1; c->body2 = pp->b2; c->geom1 = pp->g1; c->geom2 = pp->g2;
}

/* sphere (centre cs, radius r) against an axis-aligned-in-its-frame box.
 * Normal points from the sphere (geom1) to the box (geom2). Returns 1 if within margin. */
static int sphere_box(const double* cs, double r, const double* bpos, const double* bmat, const double* bsize,
                      double margin, double* dist, double* pos, double* normal) {
  double rel[3], c[3], q[3];
  sub3(rel, cs, bpos);
  mulmat3Tvec(c, bmat, rel);
  int inside = 1;
  for (int k = 0; k < 3; k++) {
    q[k] = c[k];
    if (q[k] > bsize[k]) { q[k] = bsize[k]; inside = 0; }
    if (q[k] < -bsize[k]) { q[k] = -bsize[k]; inside = 0; }
  }
  double nl[3], dd;
  if (!inside) {
    double v[3];
    sub3(v, q, c);
    dd = norm3(v);
    if (dd - r > margin) return 0;
    nl[0] = v[0] / dd; nl[1] = v[1] / dd; nl[2] = v[2] / dd;
    dd -= r;
  } else { /* centre inside the box: push out through the nearest face */
    int kbest = 0; double best = 1e30;
    for (int k = 0; k < 3; k++) { double e = bsize[k] - fabs(c[k]); if (e < best) { best = e; kbest = k; } }
    nl[0] = nl[1] = nl[2] = 0;
    nl[kbest] = c[kbest] >= 0 ? -1.0 : 1.0;
    dd = -best - r;
  }
  mulmat3vec(normal, bmat, nl);
  *dist = dd;
  pos[0] = cs[0] + normal[0] * (r + 0.5 * dd);
  pos[1] = cs[1] + normal[1] * (r + 0.5 * dd);
  pos[2] = cs[2] + normal[2] * (r + 0.5 * dd);
  return dd <= margin;
}

/* capsule (geom1) vs box (geom2): restatement of MuJoCo's mjc_CapsuleBox (engine_collision_box.c; the library is not in
 * /root/reference, DESIGN.md section 5 says what is followed from the published source and what is reconstructed).
 *
 * Followed: (1) the capsule's axis segment is brought into the box frame; (2) the box feature closest to the segment is
 * searched in MuJoCo's order — the two segment ends against the box faces (an end with at most one coordinate outside the
 * box), then the 12 box edges against the segment by clamped line-line distance, a later candidate winning only when it
 * is closer by more than MINVAL; the result is encoded as in MuJoCo: cltype -3 / -1 = a face with segment end -1 / +1,
 * cltype = 3 * s1 + s2 for an edge with s1 (edge) and s2 (segment) in {0 lower end, 1 interior, 2 upper end}; (3) at most
 * TWO contacts come out, each of them a sphere-box contact (mjraw_SphereBox: position, normal, distance, margin test) of a
 * sphere of the capsule's radius centred on the segment: the first at the closest segment point `bestsegmentpos`, the
 * second `secondpos` further along the segment, chosen by the feature type so that it is the far support point of a capsule
 * lying along a face or an edge: never beyond the segment's end, never beyond the extent of the face / edge it runs over.
 * Reconstructed (the published routine's case analysis is restated from its geometry, not line by line): the exact
 * expressions of `secondpos` in the corner / edge / face cases below; a second point that coincides with the first
 * (secondpos = 0) is not emitted. */
static void capsule_box(mzo_data* d, const pairparam* pp, const double* cpos, const double* cmat, double r, double hl,
                        const double* bpos, const double* bmat, const double* bsize) {
  double tmp1[3], tmp2[3], pos[3], axis[3], halfaxis[3], dif[3];
  const double caxis[3] = {cmat[2], cmat[5], cmat[8]};
  sub3(tmp1, cpos, bpos);
  mulmat3Tvec(pos, bmat, tmp1);      /* capsule centre in the box frame */
  mulmat3Tvec(axis, bmat, caxis);    /* capsule axis in the box frame */
  for (int k = 0; k < 3; k++) halfaxis[k] = axis[k] * hl;
  const int axisdir = (halfaxis[0] > 0 ? 1 : 0) + (halfaxis[1] > 0 ? 2 : 0) + (halfaxis[2] > 0 ? 4 : 0);
  const double bestdistmax = pp->margin + 2.0 * (r + hl + bsize[0] + bsize[1] + bsize[2]);
  double bestdist = bestdistmax * bestdistmax, bestsegmentpos = 0.0, bestboxpos = 0.0, secondpos = -4.0;
  int cltype = -4, clface = -1, clcorner = 0, cledge = -1;
  /* segment ends against the faces */
  for (int i = -1; i <= 1; i += 2) {
    int nout = 0, face = -1;
    for (int k = 0; k < 3; k++) {
      tmp1[k] = pos[k] + halfaxis[k] * i;
      tmp2[k] = tmp1[k];
      if (tmp1[k] < -bsize[k]) { nout++; face = k; tmp1[k] = -bsize[k]; }
      else if (tmp1[k] > bsize[k]) { nout++; face = k; tmp1[k] = bsize[k]; }
    }
    if (nout > 1) continue;  /* closest box feature of this end is an edge or a corner: found by the edge loop */
    sub3(dif, tmp1, tmp2);
    const double dist = dot3(dif, dif);
    if (dist < bestdist) { bestdist = dist; bestsegmentpos = i; cltype = -2 + i; clface = face; }
  }
  /* the 12 edges: edge j of the corner i whose bit j is clear (the edge runs from corner i to corner i + (1 << j)) */
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 3; j++) {
      if (i & (1 << j)) continue;
      double mid[3] = {(i & 1 ? 1 : -1) * bsize[0], (i & 2 ? 1 : -1) * bsize[1], (i & 4 ? 1 : -1) * bsize[2]};
      mid[j] = 0.0;  /* middle of the edge; edge point = mid + x1 * bsize[j] * e_j, segment point = pos + x2 * halfaxis */
      sub3(dif, mid, pos);
      const double u = -bsize[j] * dif[j], v = dot3(halfaxis, dif);
      const double ma = bsize[j] * bsize[j], mb = -bsize[j] * halfaxis[j], mc = hl * hl;
      const double det = ma * mc - mb * mb;
      if (fabs(det) < MINVAL) continue;  /* parallel: the ends already took part in the face test / other edges */
      double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
      int s1 = 1, s2 = 1;
      if (x1 > 1) { x1 = 1; s1 = 2; x2 = (v - mb) / mc; }
      else if (x1 < -1) { x1 = -1; s1 = 0; x2 = (v + mb) / mc; }
      if (x2 > 1) { x2 = 1; s2 = 2; x1 = (u - mb) / ma; if (x1 > 1) { x1 = 1; s1 = 2; } else if (x1 < -1) { x1 = -1; s1 = 0; } else s1 = 1; }
      else if (x2 < -1) { x2 = -1; s2 = 0; x1 = (u + mb) / ma; if (x1 > 1) { x1 = 1; s1 = 2; } else if (x1 < -1) { x1 = -1; s1 = 0; } else s1 = 1; }
      for (int k = 0; k < 3; k++) dif[k] = mid[k] - (pos[k] + halfaxis[k] * x2);
      dif[j] += bsize[j] * x1;
      const double dist = dot3(dif, dif);
      if (dist < bestdist - MINVAL) {
        bestdist = dist; bestsegmentpos = x2; bestboxpos = x1;
        cltype = 3 * s1 + s2;
        clcorner = i + (s1 == 2 ? (1 << j) : 0);
        cledge = j;
      }
    }
  if (cltype == -4) return;
  if (cltype >= 0 && cltype / 3 != 1) {
    /* closest to a CORNER of the box.  c1 = the coordinates in which the segment direction disagrees with the corner's
     * octant: none or all three = pointing at / away from the corner (no second contact); otherwise, along +-halfaxis
     * (mul), exactly one coordinate `ax` runs into the box's extent and the other two run out of it */
    int c1 = axisdir ^ clcorner;
    if (c1 != 0 && c1 != 7) {
      double mul = 1.0;
      if (!(c1 == 1 || c1 == 2 || c1 == 4)) { mul = -1.0; c1 = 7 - c1; }
      const int ax = c1 == 1 ? 0 : (c1 == 2 ? 1 : 2), ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (axis[ax] * axis[ax] > 0.5) { /* lying along the box edge through this corner: go on in direction mul */
        secondpos = mul * fmin(1.0 - mul * bestsegmentpos, 2.0 * bsize[ax] / fabs(halfaxis[ax]));
      } else {                         /* lying across the face: go back in direction -mul, over the face */
        double m = fmin(2.0 * bsize[ax1] / fabs(halfaxis[ax1]), 2.0 * bsize[ax2] / fabs(halfaxis[ax2]));
        secondpos = -mul * fmin(1.0 + mul * bestsegmentpos, m);
      }
    }
  } else if (cltype >= 0) {
    /* closest to the interior of an EDGE (axis cledge).  Masked to the two axes across the edge, c1 has one bit when the
     * segment passes the edge tangentially (over one face on either side): second contact over the face the segment makes
     * the smaller angle with; no bit or both bits = pointing at / away from the edge (T configuration): none */
    int c1 = (axisdir ^ clcorner) & (7 - (1 << cledge));
    if (c1 == 1 || c1 == 2 || c1 == 4) {
      const int ax = cledge;
      int ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;
      if (fabs(axis[ax1]) > fabs(axis[ax2])) { int t = ax1; ax1 = ax2; ax2 = t; }  /* ax1: normal of the face, ax2: across it */
      const double mul = (c1 & (1 << ax2)) ? 1.0 : -1.0;  /* direction along the segment that runs in over the face */
      double sp = 1.0 - mul * bestsegmentpos;                         /* to the segment's end */
      sp = fmin(sp, 2.0 * bsize[ax2] / fabs(halfaxis[ax2]));          /* to the far side of the face */
      const double e2 = (mul * halfaxis[ax] > 0) ? 1.0 - bestboxpos : 1.0 + bestboxpos;
      sp = fmin(sp, bsize[ax] * e2 / fabs(halfaxis[ax]));             /* to the end of the edge it runs along */
      secondpos = mul * sp;
    }
  } else {
    /* closest to a FACE with a segment end: the second point is the other end, pulled in so that it stays over the face */
    double travel = -2.0 * bestsegmentpos, frac = 1.0;
    for (int k = 0; k < 3; k++) {
      if (k == clface) continue;
      const double p0 = pos[k] + halfaxis[k] * bestsegmentpos, v = halfaxis[k] * travel;
      if (v > 0 && p0 + v > bsize[k]) frac = fmin(frac, (bsize[k] - p0) / v);
      if (v < 0 && p0 + v < -bsize[k]) frac = fmin(frac, (-bsize[k] - p0) / v);
    }
    secondpos = travel * fmax(frac, 0.0);
  }
  /* the contacts: spheres on the segment against the box */
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1 && !(secondpos > -3.0 && fabs(secondpos) > 1e-12)) break;
    const double t = bestsegmentpos + (pass ? secondpos : 0.0);
    double ctr[3], dist, cp[3], nrm[3];
    for (int k = 0; k < 3; k++) ctr[k] = cpos[k] + caxis[k] * hl * t;
    if (sphere_box(ctr, r, bpos, bmat, bsize, pp->margin, &dist, cp, nrm)) add_contact(d, pp, dist, cp, nrm, NULL);
  }
}

/* box (geom1) vs box (geom2): restatement of MuJoCo's mjc_BoxBox (engine_collision_box.c), face case.
 *
 * Followed: separating-axis search over the 15 axes in MuJoCo's order — per axis index i the face normal of box 1, then
 * that of box 2, a later axis winning only when its penetration is strictly smaller; then the 9 edge-edge axes, which win
 * only when clearly smaller (MuJoCo biases the comparison towards the face axes); any axis with penetration < -margin separates the pair (no contact).  For
 * a face axis the reference face is the winning face, the incident face is the face of the other box most anti-parallel
 * to it, and the contact candidates are the vertices of the intersection of the incident rectangle (projected along the
 * normal) with the reference rectangle: incident corners inside the reference rectangle, crossings of incident edges
 * with the reference rectangle's border lines, reference corners inside the incident rectangle (taken on the incident
 * plane).  Each candidate carries its own distance to the reference face; candidates beyond the margin are dropped; a
 * contact sits midway between the candidate and the reference face; the normal is the reference normal oriented from
 * geom1 to geom2; at most 8 contacts.
 * Reconstructed: all inside / crossing tests are inclusive (a candidate ON a border line of the reference rectangle counts:
 * the movable blocks have exactly the walls' z extent), a candidate that coincides with an earlier one is emitted once
 * (MuJoCo's enumeration can list such a vertex twice), and — the one deliberate deviation — an intersection without area makes
 * no contact (see MZO_BOX_MINOVERLAP below).  The edge-edge case (no registered maze can reach it: every box here is rotated about z only, where
 * each edge-edge axis coincides with a face axis and loses the tie) yields one contact midway between the closest points of
 * the two edges. */
static void box_box(mzo_data* d, const pairparam* pp, const double* pos1, const double* mat1, const double* size1,
                    const double* pos2, const double* mat2, const double* size2) {
  double rot[9], rotabs[9], pos21[3], pos12[3], tmp[3], plen1[3], plen2[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {  /* rot[i][j] = axis i of box 1 . axis j of box 2 */
      rot[3 * i + j] = mat1[i] * mat2[j] + mat1[3 + i] * mat2[3 + j] + mat1[6 + i] * mat2[6 + j];
      rotabs[3 * i + j] = fabs(rot[3 * i + j]);
    }
  sub3(tmp, pos2, pos1); mulmat3Tvec(pos21, mat1, tmp);
  sub3(tmp, pos1, pos2); mulmat3Tvec(pos12, mat2, tmp);
  for (int i = 0; i < 3; i++) {
    plen2[i] = rotabs[3 * i] * size2[0] + rotabs[3 * i + 1] * size2[1] + rotabs[3 * i + 2] * size2[2];
    plen1[i] = rotabs[i] * size1[0] + rotabs[3 + i] * size1[1] + rotabs[6 + i] * size1[2];
  }
  const double margin = pp->margin;
  double penetration = margin;
  for (int i = 0; i < 3; i++) penetration += 3.0 * (size1[i] + size2[i]);
  int code = -1;
  for (int i = 0; i < 3; i++) {
    const double c1 = -fabs(pos21[i]) + size1[i] + plen2[i], c2 = -fabs(pos12[i]) + size2[i] + plen1[i];
    if (c1 < -margin || c2 < -margin) return;
    if (c1 < penetration) { penetration = c1; code = i + 3 * (pos21[i] < 0) ; }
    if (c2 < penetration) { penetration = c2; code = 6 + i + 3 * (pos12[i] < 0); }
  }
  double en[3] = {0, 0, 0};  /* edge-edge axis (frame of box 1, pointing from box 1 to box 2) */
  int ei = -1, ej = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double ei3[3] = {i == 0, i == 1, i == 2}, aj[3] = {rot[j], rot[3 + j], rot[6 + j]};
      double cr[3];
      cross3(cr, ei3, aj);
      const double len = norm3(cr);
      if (len < 1e-10) continue;
      for (int k = 0; k < 3; k++) cr[k] /= len;
      double r1 = 0, r2 = 0;
      for (int k = 0; k < 3; k++) {
        r1 += size1[k] * fabs(cr[k]);
        r2 += size2[k] * fabs(cr[0] * rot[k] + cr[1] * rot[3 + k] + cr[2] * rot[6 + k]);
      }
      const double sep = dot3(pos21, cr), c3 = r1 + r2 - fabs(sep);
      if (c3 < -margin) return;
      /* wins only when clearly better than the best face axis (whatever the sign of the penetration): an edge-edge axis
       * that coincides with a face axis (parallel boxes, rotation about one axis) must lose the tie, round-off included */
      if (c3 < penetration - 1e-10 * (fabs(penetration) + r1 + r2)) {
        penetration = c3; code = 12 + 3 * i + j; ei = i; ej = j;
        for (int k = 0; k < 3; k++) en[k] = sep >= 0 ? cr[k] : -cr[k];
      }
    }
  if (code < 0) return;
  if (code >= 12) {
    /* edge-edge: the edge of box 1 along axis ei that is farthest along en, the edge of box 2 along axis ej farthest against it */
    double p1[3], p2[3], d1[3] = {ei == 0, ei == 1, ei == 2}, d2[3] = {rot[ej], rot[3 + ej], rot[6 + ej]};
    for (int k = 0; k < 3; k++) p1[k] = k == ei ? 0.0 : (en[k] >= 0 ? size1[k] : -size1[k]);
    cpy3(p2, pos21);
    for (int k = 0; k < 3; k++) {
      if (k == ej) continue;
      const double ak[3] = {rot[k], rot[3 + k], rot[6 + k]};
      addscl3(p2, ak, dot3(ak, en) >= 0 ? -size2[k] : size2[k]);
    }
    /* closest points of the two lines p1 + s d1, p2 + t d2 */
    double w[3]; sub3(w, p1, p2);
    const double b = dot3(d1, d2), dd = dot3(d1, w), e = dot3(d2, w), den = 1.0 - b * b;
    const double sc = den > 1e-12 ? (b * e - dd) / den : 0.0, tc = den > 1e-12 ? (e - b * dd) / den : 0.0;
    double q1[3], q2[3], mid[3], nw[3], pw[3];
    for (int k = 0; k < 3; k++) { q1[k] = p1[k] + sc * d1[k]; q2[k] = p2[k] + tc * d2[k]; mid[k] = 0.5 * (q1[k] + q2[k]); }
    mulmat3vec(nw, mat1, en); mulmat3vec(pw, mat1, mid); add3(pw, pw, pos1);
    add_contact(d, pp, -penetration, pw, nw, NULL);
    return;
  }
  /* face case, worked in the frame of the reference box A */
  const int fromB = code >= 6, a = code % 3;
  const double* posA = fromB ? pos2 : pos1; const double* matA = fromB ? mat2 : mat1;
  const double* sA = fromB ? size2 : size1; const double* sB = fromB ? size1 : size2;
  const double* pBA = fromB ? pos12 : pos21;
  double R[9];  /* R[k][j] = axis k of A . axis j of B */
  for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) R[3 * k + j] = fromB ? rot[3 * j + k] : rot[3 * k + j];
  const double sg = pBA[a] < 0 ? -1.0 : 1.0;  /* the reference normal sg * e_a points from A to B */
  int b = 0;
  for (int j = 1; j < 3; j++) if (fabs(R[3 * a + j]) > fabs(R[3 * a + b])) b = j;
  const double sb = R[3 * a + b] * sg > 0 ? -1.0 : 1.0;  /* incident face: the one of B whose outward normal opposes the reference normal */
  const int b1 = (b + 1) % 3, b2 = (b + 2) % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
  double cf[3], u[3], v[3];
  for (int k = 0; k < 3; k++) {
    cf[k] = pBA[k] + sb * sB[b] * R[3 * k + b];
    u[k] = sB[b1] * R[3 * k + b1];
    v[k] = sB[b2] * R[3 * k + b2];
  }
  double cand[24][3];  /* 4 incident corners + 4 edges x 2 border lines x 2 sides... at most 4 + 16 + 4 */
  int nc = 0;
  const double tol = 1e-12 * (1.0 + sA[a1] + sA[a2]);
  /* incident corners inside the reference rectangle */
  for (int c = 0; c < 4; c++) {
    const double su = (c & 1) ? 1.0 : -1.0, sv = (c & 2) ? 1.0 : -1.0;
    double p[3];
    for (int k = 0; k < 3; k++) p[k] = cf[k] + su * u[k] + sv * v[k];
    if (fabs(p[a1]) <= sA[a1] + tol && fabs(p[a2]) <= sA[a2] + tol) { cpy3(cand[nc], p); nc++; }
  }
  /* incident edges against the border lines of the reference rectangle */
  for (int e = 0; e < 4; e++) {
    double p[3], q[3];  /* edge = p + t q, t in [-1, 1] */
    for (int k = 0; k < 3; k++) {
      if (e < 2) { p[k] = cf[k] + (e ? v[k] : -v[k]); q[k] = u[k]; }
      else { p[k] = cf[k] + (e == 3 ? u[k] : -u[k]); q[k] = v[k]; }
    }
    for (int w = 0; w < 2; w++) {
      const int c = w ? a2 : a1, o = w ? a1 : a2;
      if (fabs(q[c]) < MINVAL) continue;
      for (int side = -1; side <= 1; side += 2) {
        const double t = (side * sA[c] - p[c]) / q[c];
        if (t < -1.0 || t > 1.0) continue;
        if (fabs(p[o] + t * q[o]) > sA[o] + tol) continue;
        for (int k = 0; k < 3; k++) cand[nc][k] = p[k] + t * q[k];
        nc++;
      }
    }
  }
  /* reference corners inside the incident rectangle, on the incident plane */
  {
    const double det = u[a1] * v[a2] - u[a2] * v[a1];
    if (fabs(det) > MINVAL)
      for (int c = 0; c < 4; c++) {
        const double x = ((c & 1) ? sA[a1] : -sA[a1]) - cf[a1], y = ((c & 2) ? sA[a2] : -sA[a2]) - cf[a2];
        const double al = (x * v[a2] - y * v[a1]) / det, be = (u[a1] * y - u[a2] * x) / det;
        if (fabs(al) <= 1.0 + 1e-12 && fabs(be) <= 1.0 + 1e-12) {
          for (int k = 0; k < 3; k++) cand[nc][k] = cf[k] + al * u[k] + be * v[k];
          nc++;
        }
      }
  }
  int emitted = 0;
  const double dtol = 1e-9 * (1.0 + sA[a1] + sA[a2]);  /* candidates closer than this (max norm) are one vertex */
  {
    /* The faces must overlap by a positive area: an intersection that has collapsed to a segment or a point (narrower than
     * MZO_BOX_MINOVERLAP in a face direction) makes no contact.  This is the one deliberate deviation from the inclusive
     * enumeration [ASSUME-12]: grid-aligned boxes that share only a border line — every movable block at its spawn position
     * against the diagonal wall cells — would otherwise sit on a knife edge (zero-width contact strips that appear and
     * vanish with the sign of a 1e-16 displacement), in the reference as much as here. */
    double lo1 = 1e300, hi1 = -1e300, lo2 = 1e300, hi2 = -1e300;
    for (int c = 0; c < nc; c++) {
      lo1 = fmin(lo1, cand[c][a1]); hi1 = fmax(hi1, cand[c][a1]);
      lo2 = fmin(lo2, cand[c][a2]); hi2 = fmax(hi2, cand[c][a2]);
    }
    if (nc == 0 || hi1 - lo1 <= MZO_BOX_MINOVERLAP || hi2 - lo2 <= MZO_BOX_MINOVERLAP) return;
  }
  for (int c = 0; c < nc && emitted < 8; c++) {
    int dup = 0;
    for (int e = 0; e < c && !dup; e++)
      if (fabs(cand[c][0] - cand[e][0]) <= dtol && fabs(cand[c][1] - cand[e][1]) <= dtol && fabs(cand[c][2] - cand[e][2]) <= dtol) dup = 1;
    if (dup) continue;
    const double dist = sg * cand[c][a] - sA[a];
    if (dist > margin) continue;
    double pl[3], pw[3], nl[3] = {0, 0, 0}, nw[3];
    cpy3(pl, cand[c]);
    pl[a] -= sg * 0.5 * dist;
    nl[a] = fromB ? -sg : sg;  /* reported from geom1 to geom2 */
    mulmat3vec(pw, matA, pl); add3(pw, pw, posA);
    mulmat3vec(nw, matA, nl);
    add_contact(d, pp, dist, pw, nw, NULL);
    emitted++;
  }
}

static void mix_params(pairparam* pp, double m1, double m2, double g1, double g2, const double* f1, const double* f2,
                       const double* sr1, const double* sr2, const double* si1, const double* si2, int cd1, int cd2) {
  pp->margin = fmax(m1, m2);
  pp->gap = fmax(g1, g2);
  pp->mu = fmax(f1[0], f2[0]);
  for (int k = 0; k < 2; k++) pp->solref[k] = 0.5 * (sr1[k] + sr2[k]); /* [ASSUME-4] equal solmix / priority */
  for (int k = 0; k < 5; k++) pp->solimp[k] = 0.5 * (si1[k] + si2[k]);
  pp->condim = cd1 > cd2 ? cd1 : cd2;
}

static void collide_plane(const mz_model* m, mzo_data* d, int gp, int g) {
  pairparam pp;
  mix_params(&pp, m->geom_margin[gp], m->geom_margin[g], m->geom_gap[gp], m->geom_gap[g], m->geom_friction[gp],
             m->geom_friction[g], m->geom_solref[gp], m->geom_solref[g], m->geom_solimp[gp], m->geom_solimp[g],
             m->geom_condim[gp], m->geom_condim[g]);
  pp.b1 = m->geom_bodyid[gp]; pp.b2 = m->geom_bodyid[g]; pp.g1 = gp; pp.g2 = g;
  const double* pm = d->geom_xmat[gp];
  double n[3] = {pm[2], pm[5], pm[8]}, rel[3];
  if (m->geom_type[g] == MZ_GEOM_SPHERE) {
    double r = m->geom_size[g][0];
    sub3(rel, d->geom_xpos[g], d->geom_xpos[gp]);
    double dist = dot3(rel, n) - r;
    if (dist > pp.margin) return;
    double pos[3];
    for (int k = 0; k < 3; k++) pos[k] = d->geom_xpos[g][k] - n[k] * (r + 0.5 * dist);
    add_contact(d, &pp, dist, pos, n, NULL);
  } else if (m->geom_type[g] == MZ_GEOM_CAPSULE) {
    double r = m->geom_size[g][0], hl = m->geom_size[g][1];
    const double* cm = d->geom_xmat[g];
    double axis[3] = {cm[2], cm[5], cm[8]};
    for (int s = 0; s < 2; s++) { /* [ASSUME-5] end +axis first, then -axis; tangent hint = capsule axis */
      double sg = s == 0 ? 1.0 : -1.0, e[3];
      for (int k = 0; k < 3; k++) e[k] = d->geom_xpos[g][k] + sg * axis[k] * hl;
      sub3(rel, e, d->geom_xpos[gp]);
      double dist = dot3(rel, n) - r;
      if (dist > pp.margin) continue;
      double pos[3];
      for (int k = 0; k < 3; k++) pos[k] = e[k] - n[k] * (r + 0.5 * dist);
      add_contact(d, &pp, dist, pos, n, axis);
    }
  } else if (m->geom_type[g] == MZ_GEOM_BOX) {
    /* plane-box (mjc_PlaneBox): the corners in index order (bit 0 x, bit 1 y, bit 2 z); a corner is skipped when it lies
     * beyond the margin or on the far side of the box centre (its offset along the plane normal is positive); at most 4 */
    const double* bm = d->geom_xmat[g];
    const double* sz = m->geom_size[g];
    sub3(rel, d->geom_xpos[g], d->geom_xpos[gp]);
    const double cdist = dot3(rel, n);
    int cnt = 0;
    for (int cidx = 0; cidx < 8 && cnt < 4; cidx++) {
      double loc[3] = {(cidx & 1 ? 1 : -1) * sz[0], (cidx & 2 ? 1 : -1) * sz[1], (cidx & 4 ? 1 : -1) * sz[2]}, w[3], e[3];
      mulmat3vec(w, bm, loc);
      const double ldist = dot3(n, w);
      if (cdist + ldist > pp.margin || ldist > 0) continue;
      const double dist = cdist + ldist;
      add3(e, d->geom_xpos[g], w);
      double pos[3];
      for (int k = 0; k < 3; k++) pos[k] = e[k] - n[k] * (0.5 * dist);
      add_contact(d, &pp, dist, pos, n, NULL);
      cnt++;
    }
  }
}

static void collide_walls(const mz_model* m, mzo_data* d, int g) {
  /* implicit maze boxes (world body): only cells overlapping the geom's bounding square can touch */
  if (!((m->geom_contype[g] & m->wall_conaffinity) || (m->wall_contype & m->geom_conaffinity[g]))) return;
  pairparam pp;
  mix_params(&pp, m->geom_margin[g], m->wall_margin, m->geom_gap[g], m->wall_gap, m->geom_friction[g], m->wall_friction,
             m->geom_solref[g], m->wall_solref, m->geom_solimp[g], m->wall_solimp, m->geom_condim[g], m->wall_condim);
  pp.b1 = m->geom_bodyid[g]; pp.b2 = 0; pp.g1 = g; pp.g2 = -1;
  double s = m->maze_scale, reach = m->geom_rbound[g] + pp.margin;
  const double* gp = d->geom_xpos[g];
  if (gp[2] - reach > m->wall_center_z + m->wall_half_z) return;
  if (!m->elevated && gp[2] + reach < m->wall_center_z - m->wall_half_z) return;
  int j0 = (int)floor((gp[0] - reach + m->torso_x) / s + 0.5), j1 = (int)floor((gp[0] + reach + m->torso_x) / s + 0.5);
  int i0 = (int)floor((gp[1] - reach + m->torso_y) / s + 0.5), i1 = (int)floor((gp[1] + reach + m->torso_y) / s + 0.5);
  static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double bsize[3] = {m->wall_half_xy, m->wall_half_xy, m->wall_half_z};
  for (int i = i0; i <= i1; i++)
    for (int j = j0; j <= j1; j++) {
      if (i < 0 || j < 0 || i >= m->grid_rows || j >= m->grid_cols) continue;
      /* per cell, in the order the reference emits the geoms (maze_env.py:124-152): the platform of an elevated maze
       * (every cell but the chasms; z from 0 to height_offset), then the wall block standing on it */
      for (int layer = 0; layer < 2; layer++) {
        if (layer == 0 && !(m->elevated && m->grid[i][j] != MZ_CELL_CHASM)) continue;
        if (layer == 1 && m->grid[i][j] != MZ_CELL_BLOCK) continue;
        double bpos[3] = {j * s - m->torso_x, i * s - m->torso_y, layer == 0 ? m->wall_half_z : m->wall_center_z};
        if (gp[2] - reach > bpos[2] + m->wall_half_z || gp[2] + reach < bpos[2] - m->wall_half_z) continue;
        if (m->geom_type[g] == MZ_GEOM_SPHERE) {
          double dist, pos[3], nrm[3];
          if (sphere_box(gp, m->geom_size[g][0], bpos, ident, bsize, pp.margin, &dist, pos, nrm))
            add_contact(d, &pp, dist, pos, nrm, NULL);
        } else if (m->geom_type[g] == MZ_GEOM_CAPSULE) {
          capsule_box(d, &pp, gp, d->geom_xmat[g], m->geom_size[g][0], m->geom_size[g][1], bpos, ident, bsize);
        } else if (m->geom_type[g] == MZ_GEOM_BOX) {
          /* wall geoms precede the robot's and the movable bodies' geoms in MuJoCo's geom order: geom1 = wall, geom2 = box */
          pairparam q = pp;
          q.b1 = 0; q.b2 = m->geom_bodyid[g]; q.g1 = -1; q.g2 = g;
          box_box(d, &q, bpos, ident, bsize, gp, d->geom_xmat[g], m->geom_size[g]);
        } else {
          d->status |= MZO_STATUS_UNSUPPORTED_PAIR; /* generally rotated box vs box: not restated */
        }
      }
    }
}

/* two spheres (mjraw_SphereSphere): normal from the first to the second; returns the number of contacts added */
static int sphere_pair(mzo_data* d, const pairparam* pp, const double* c1, double r1, const double* c2, double r2) {
  double dv[3], pos[3], nrm[3];
  sub3(dv, c2, c1);
  const double cd = norm3(dv), dist = cd - r1 - r2;
  if (dist > pp->margin) return 0;
  if (cd < MINVAL) { nrm[0] = 1.0; nrm[1] = 0.0; nrm[2] = 0.0; }
  else { nrm[0] = dv[0] / cd; nrm[1] = dv[1] / cd; nrm[2] = dv[2] / cd; }
  for (int k = 0; k < 3; k++) pos[k] = c1[k] + nrm[k] * (r1 + 0.5 * dist);
  add_contact(d, pp, dist, pos, nrm, NULL);
  return 1;
}

/* capsule vs capsule (MuJoCo engine_collision_primitive.c mjraw_CapsuleCapsule, restated): the nearest points of the two axis segments
 * — the unconstrained minimiser of |p1 + x1 a1 - p2 - x2 a2|^2, clamped to [-1, 1] one coordinate after the other — then sphere-sphere
 * between them; parallel axes (|det| < mjMINVAL): the segment ends, up to two contacts.  a1, a2: axes scaled by the half lengths. */
static void capsule_capsule(mzo_data* d, const pairparam* pp, const double* pos1, const double* mat1, double r1, double hl1,
                            const double* pos2, const double* mat2, double r2, double hl2) {
  const double a1[3] = {mat1[2] * hl1, mat1[5] * hl1, mat1[8] * hl1}, a2[3] = {mat2[2] * hl2, mat2[5] * hl2, mat2[8] * hl2};
  double dif[3], v1[3], v2[3];
  sub3(dif, pos1, pos2);
  const double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif), det = ma * mc - mb * mb;
#define MZO_CLAMP1(x) ((x) > 1.0 ? 1.0 : ((x) < -1.0 ? -1.0 : (x)))
#define MZO_PTS(x1, x2) do { for (int k = 0; k < 3; k++) { v1[k] = pos1[k] + a1[k] * (x1); v2[k] = pos2[k] + a2[k] * (x2); } } while (0)
  if (fabs(det) >= MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > 1.0) { x1 = 1.0; x2 = (v - mb) / mc; }
    else if (x1 < -1.0) { x1 = -1.0; x2 = (v + mb) / mc; }
    if (x2 > 1.0) { x2 = 1.0; x1 = MZO_CLAMP1((u - mb) / ma); }
    else if (x2 < -1.0) { x2 = -1.0; x1 = MZO_CLAMP1((u + mb) / ma); }
    MZO_PTS(x1, x2);
    sphere_pair(d, pp, v1, r1, v2, r2);
    return;
  }
  int n = 0;
  double x;
  x = MZO_CLAMP1((v - mb) / mc); MZO_PTS(1.0, x); n += sphere_pair(d, pp, v1, r1, v2, r2);   /* x1 = 1 */
  x = MZO_CLAMP1((v + mb) / mc); MZO_PTS(-1.0, x); n += sphere_pair(d, pp, v1, r1, v2, r2);  /* x1 = -1 */
  if (n >= 2) return;
  x = MZO_CLAMP1((u - mb) / ma); MZO_PTS(x, 1.0); n += sphere_pair(d, pp, v1, r1, v2, r2);   /* x2 = 1 */
  if (n >= 2) return;
  x = MZO_CLAMP1((u + mb) / ma); MZO_PTS(x, -1.0); sphere_pair(d, pp, v1, r1, v2, r2);       /* x2 = -1 */
#undef MZO_PTS
#undef MZO_CLAMP1
}

/* explicit non-plane pairs: robot sphere / capsule against a movable block's box, block against block */
static void collide_pair(const mz_model* m, mzo_data* d, int ga, int gb) {
  int g1 = ga, g2 = gb;
  if (m->geom_type[g1] > m->geom_type[g2]) { g1 = gb; g2 = ga; } /* MuJoCo orders a pair by geom type */
  pairparam pp;
  mix_params(&pp, m->geom_margin[g1], m->geom_margin[g2], m->geom_gap[g1], m->geom_gap[g2], m->geom_friction[g1],
             m->geom_friction[g2], m->geom_solref[g1], m->geom_solref[g2], m->geom_solimp[g1], m->geom_solimp[g2],
             m->geom_condim[g1], m->geom_condim[g2]);
  pp.b1 = m->geom_bodyid[g1]; pp.b2 = m->geom_bodyid[g2]; pp.g1 = g1; pp.g2 = g2;
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  if (t1 == MZ_GEOM_SPHERE && t2 == MZ_GEOM_SPHERE) { /* the Point's body against an object ball: normal from geom1 to geom2 */
    double dv[3], pos[3], nrm[3];
    sub3(dv, d->geom_xpos[g2], d->geom_xpos[g1]);
    double cd = norm3(dv), r1 = m->geom_size[g1][0], r2 = m->geom_size[g2][0], dist = cd - r1 - r2;
    if (dist > pp.margin) return;
    if (cd < MINVAL) { nrm[0] = 1.0; nrm[1] = 0.0; nrm[2] = 0.0; }
    else { nrm[0] = dv[0] / cd; nrm[1] = dv[1] / cd; nrm[2] = dv[2] / cd; }
    for (int k = 0; k < 3; k++) pos[k] = d->geom_xpos[g1][k] + nrm[k] * (r1 + 0.5 * dist);
    add_contact(d, &pp, dist, pos, nrm, NULL);
  } else if (t1 == MZ_GEOM_SPHERE && t2 == MZ_GEOM_CAPSULE) {
    /* an object ball (free joint) against a leg capsule of the ant: the point of the capsule's axis segment nearest to the
     * sphere centre, then sphere-sphere; normal from geom1 (the sphere) to geom2 (mjc_SphereCapsule) */
    const double* cm = d->geom_xmat[g2];
    double axis[3] = {cm[2], cm[5], cm[8]}, vec[3], pt[3], dv[3], pos[3], nrm[3];
    sub3(vec, d->geom_xpos[g1], d->geom_xpos[g2]);
    double hl = m->geom_size[g2][1], x = dot3(axis, vec);
    x = x < -hl ? -hl : (x > hl ? hl : x);
    for (int k = 0; k < 3; k++) pt[k] = d->geom_xpos[g2][k] + axis[k] * x;
    sub3(dv, pt, d->geom_xpos[g1]);
    double cd = norm3(dv), r1 = m->geom_size[g1][0], r2 = m->geom_size[g2][0], dist = cd - r1 - r2;
    if (dist > pp.margin) return;
    if (cd < MINVAL) { nrm[0] = 1.0; nrm[1] = 0.0; nrm[2] = 0.0; }
    else { nrm[0] = dv[0] / cd; nrm[1] = dv[1] / cd; nrm[2] = dv[2] / cd; }
    for (int k = 0; k < 3; k++) pos[k] = d->geom_xpos[g1][k] + nrm[k] * (r1 + 0.5 * dist);
    add_contact(d, &pp, dist, pos, nrm, NULL);
  } else if (t1 == MZ_GEOM_CAPSULE && t2 == MZ_GEOM_CAPSULE) { /* a robot's own limbs against each other, same type: geom1 = the lower id */
    capsule_capsule(d, &pp, d->geom_xpos[g1], d->geom_xmat[g1], m->geom_size[g1][0], m->geom_size[g1][1], d->geom_xpos[g2], d->geom_xmat[g2],
                    m->geom_size[g2][0], m->geom_size[g2][1]);
  } else if (t2 == MZ_GEOM_BOX && t1 == MZ_GEOM_SPHERE) {
    double dist, pos[3], nrm[3];
    if (sphere_box(d->geom_xpos[g1], m->geom_size[g1][0], d->geom_xpos[g2], d->geom_xmat[g2], m->geom_size[g2], pp.margin, &dist, pos, nrm))
      add_contact(d, &pp, dist, pos, nrm, NULL);
  } else if (t2 == MZ_GEOM_BOX && t1 == MZ_GEOM_CAPSULE) {
    capsule_box(d, &pp, d->geom_xpos[g1], d->geom_xmat[g1], m->geom_size[g1][0], m->geom_size[g1][1], d->geom_xpos[g2],
                d->geom_xmat[g2], m->geom_size[g2]);
  } else if (t1 == MZ_GEOM_BOX && t2 == MZ_GEOM_BOX) {  /* same type: geom1 = the lower geom id (the Point's arrow before a movable block) */
    box_box(d, &pp, d->geom_xpos[g1], d->geom_xmat[g1], m->geom_size[g1], d->geom_xpos[g2], d->geom_xmat[g2], m->geom_size[g2]);
  } else {
    d->status |= MZO_STATUS_UNSUPPORTED_PAIR;
  }
}

static void collision(const mz_model* m, mzo_data* d) {
  d->ncon = 0;
  if (m->collision_predefined) return; /* swimmer.xml:3 */
  for (int g1 = 0; g1 < m->ngeom; g1++)
    for (int g2 = g1 + 1; g2 < m->ngeom; g2++) {
      int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
      if (b1 == b2) continue;
      if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) continue;
      /* parent-child filter does not apply when the parent is the world body */
      if (b1 != 0 && b2 != 0 && (m->body_parent[b1] == b2 || m->body_parent[b2] == b1)) continue;
      if (m->geom_type[g1] == MZ_GEOM_PLANE) collide_plane(m, d, g1, g2);
      else collide_pair(m, d, g1, g2);
    }
  for (int g = 1; g < m->ngeom; g++)
    if (m->geom_bodyid[g] != 0) collide_walls(m, d, g);
}

/* Narrow-phase probes (tests/test_narrowphase_probes.py, tests/test_mujoco_crosscheck.py): one routine on one hand-made pose.
 * kind 0 capsule (geom1: pos, mat, size = radius, half length) vs box (geom2); 1 box vs box; 2 sphere (size[0]) vs box; 3 capsule vs capsule.
 * out: per contact 7 doubles dist | pos | normal (geom1 -> geom2); returns the contact count. */
int mzo_probe_pair(int kind, const double* pos1, const double* mat1, const double* size1, const double* pos2, const double* mat2,
                   const double* size2, double margin, int max_con, double* out) {
  static const double sr[2] = {0.02, 1.0}, si[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
  mzo_data* d = (mzo_data*)calloc(1, sizeof(mzo_data));
  pairparam pp;
  memset(&pp, 0, sizeof(pp));
  pp.margin = margin; pp.mu = 1.0; pp.condim = 3;
  memcpy(pp.solref, sr, sizeof(sr)); memcpy(pp.solimp, si, sizeof(si));
  if (kind == 0) capsule_box(d, &pp, pos1, mat1, size1[0], size1[1], pos2, mat2, size2);
  else if (kind == 3) capsule_capsule(d, &pp, pos1, mat1, size1[0], size1[1], pos2, mat2, size2[0], size2[1]);
  else if (kind == 1) box_box(d, &pp, pos1, mat1, size1, pos2, mat2, size2);
  else {
    double dist, cp[3], nrm[3];
    if (sphere_box(pos1, size1[0], pos2, mat2, size2, margin, &dist, cp, nrm)) add_contact(d, &pp, dist, cp, nrm, NULL);
  }
  const int n = d->ncon;
  for (int c = 0; c < n && c < max_con; c++) {
    out[7 * c] = d->con[c].dist;
    for (int k = 0; k < 3; k++) { out[7 * c + 1 + k] = d->con[c].pos[k]; out[7 * c + 4 + k] = d->con[c].frame[k]; }
  }
  free(d);
  return n;
}

/* ------------------------------------------------------------------ constraints */
static void point_jacobian(const mz_model* m, const mzo_data* d, int body, const double* p, double jac[3][ND]) {
  for (int i = 0; i < m->nv; i++) jac[0][i] = jac[1][i] = jac[2][i] = 0.0;
  double off[3];
  sub3(off, p, d->refpoint);
  for (int b = body; b > 0; b = m->body_parent[b])
    for (int i = m->body_dofadr[b]; i >= 0 && i < m->body_dofadr[b] + m->body_dofnum[b]; i++) {
      double wx[3];
      cross3(wx, d->S[i], off);
      jac[0][i] = d->S[i][3] + wx[0]; jac[1][i] = d->S[i][4] + wx[1]; jac[2][i] = d->S[i][5] + wx[2];
    }
}

static double impedance(const double* solimp, double x) { /* x = |pos - margin| */
  double d0 = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (d0 == dmax || width <= MINVAL) return 0.5 * (d0 + dmax);
  double xn = x / width;
  if (xn >= 1.0) return dmax;
  if (xn <= 0.0) return d0;
  double y;
  if (power <= 1.0 + 1e-12) y = xn;
  else if (xn <= mid) y = pow(xn, power) / pow(mid, power - 1.0);
  else y = 1.0 - pow(1.0 - xn, power) / pow(1.0 - mid, power - 1.0);
  return d0 + y * (dmax - d0);
}

static void add_row(const mz_model* m, mzo_data* d, const double* J, double pos, double margin, double diag,
                    const double* solref, const double* solimp) {
  if (d->nefc >= MZO_MAX_EFC) { d->status |= MZ_STATUS_CONTACT_OVERFLOW; return; }
  int r = d->nefc++;
  memcpy(d->efc_J[r], J, sizeof(double) * m->nv);
  d->efc_pos[r] = pos;
  d->efc_margin[r] = margin;
  double imp = impedance(solimp, fabs(pos - margin));
  d->efc_R[r] = fmax(MINVAL, (1.0 - imp) * diag / imp);
  /* reference acceleration coefficients: solref = (timeconst, dampratio), refsafe on [ASSUME-2] */
  double tc = fmax(solref[0], 2.0 * m->timestep), dr = solref[1], dmax = solimp[1];
  d->efc_K[r] = 1.0 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
  d->efc_B[r] = 2.0 / fmax(MINVAL, dmax * tc);
  d->efc_imp[r] = imp;
}

static void make_constraints(const mz_model* m, mzo_data* d) {
  d->nefc = 0;
  int nv = m->nv;
  double J[ND];
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || (m->jnt_type[j] != MZ_JNT_HINGE && m->jnt_type[j] != MZ_JNT_SLIDE)) continue;
    double q = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side < 0 ? q - m->jnt_range[j][0] : m->jnt_range[j][1] - q;
      if (dist < m->jnt_margin[j]) {
        memset(J, 0, sizeof(J));
        J[m->jnt_dofadr[j]] = -(double)side;
        add_row(m, d, J, dist, m->jnt_margin[j], m->dof_invweight0[m->jnt_dofadr[j]], m->jnt_solref[j], m->jnt_solimp[j]);
      }
    }
  }
  /* contacts: pyramidal cone, condim 3 -> 4 rows (condim 1 -> 1 frictionless row) */
  for (int c = 0; c < d->ncon; c++) {
    mzo_contact* con = &d->con[c];
    con->efc_address = -1;
    if (!(con->dist < con->includemargin)) continue;
    double j1[3][ND], j2[3][ND], jc[3][ND];
    point_jacobian(m, d, con->body1, con->pos, j1);
    point_jacobian(m, d, con->body2, con->pos, j2);
    for (int k = 0; k < 3; k++)
      for (int i = 0; i < nv; i++)
        jc[k][i] = con->frame[3 * k] * (j2[0][i] - j1[0][i]) + con->frame[3 * k + 1] * (j2[1][i] - j1[1][i]) +
                   con->frame[3 * k + 2] * (j2[2][i] - j1[2][i]);
    double tran = m->body_invweight0[con->body1][0] + m->body_invweight0[con->body2][0];
    con->efc_address = d->nefc;
    if (con->dim == 1) {
      add_row(m, d, jc[0], con->dist, con->includemargin, tran, con->solref, con->solimp);
      continue;
    }
    int first = d->nefc;
    double mu = con->mu;
    for (int k = 1; k <= 2; k++)
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        for (int i = 0; i < nv; i++) J[i] = jc[0][i] + sgn * mu * jc[k][i];
        add_row(m, d, J, con->dist, con->includemargin, tran + mu * mu * tran, con->solref, con->solimp);
      }
    /* [ASSUME-3] pyramid edges share R = 2 mu^2 R(first edge) */
    double Rpy = 2.0 * mu * mu * d->efc_R[first];
    for (int r = first; r < d->nefc; r++) d->efc_R[r] = Rpy;
  }
  for (int r = 0; r < d->nefc; r++) {
    double vel = 0;
    for (int i = 0; i < nv; i++) vel += d->efc_J[r][i] * d->qvel[i];
    d->efc_D[r] = 1.0 / d->efc_R[r];
    d->efc_aref[r] = -d->efc_B[r] * vel - d->efc_K[r] * d->efc_imp[r] * (d->efc_pos[r] - d->efc_margin[r]);
  }
}

/* ------------------------------------------------------------------ Newton solver */
typedef struct { double a; double jar, jv, D; } lsbp;
static int cmp_bp(const void* x, const void* y) {
  double a = ((const lsbp*)x)->a, b = ((const lsbp*)y)->a;
  return (a > b) - (a < b);
}

static double eval_cost(const mz_model* m, const mzo_data* d, const double* qacc, double* jar_out) {
  int nv = m->nv;
  double cost = 0;
  for (int i = 0; i < nv; i++) {
    double s = 0;
    for (int j = 0; j < nv; j++) s += d->M[i][j] * (qacc[j] - d->qacc_smooth[j]);
    cost += 0.5 * s * (qacc[i] - d->qacc_smooth[i]);
  }
  for (int r = 0; r < d->nefc; r++) {
    double jar = -d->efc_aref[r];
    for (int i = 0; i < nv; i++) jar += d->efc_J[r][i] * qacc[i];
    if (jar_out) jar_out[r] = jar;
    if (jar < 0) cost += 0.5 * d->efc_D[r] * jar * jar;
  }
  return cost;
}

static void solve_constraints(const mz_model* m, mzo_data* d, double tol, int maxiter) {
  int nv = m->nv, ne = d->nefc;
  d->solver_iter = 0;
  if (ne == 0) { memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv); return; }
  double qacc[ND], jar[MZO_MAX_EFC], grad[ND], search[ND], Ms[ND], jv[MZO_MAX_EFC];
  /* warm start: the better of qacc_warmstart and qacc_smooth */
  double cw = eval_cost(m, d, d->qacc_warmstart, NULL), cs = eval_cost(m, d, d->qacc_smooth, NULL);
  memcpy(qacc, cw < cs ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  double cost = eval_cost(m, d, qacc, jar);
  double H[ND][ND];
  for (int it = 0; it < maxiter; it++) {
    /* gradient and Hessian on the current active set */
    for (int i = 0; i < nv; i++) {
      double s = 0;
      for (int j = 0; j < nv; j++) s += d->M[i][j] * (qacc[j] - d->qacc_smooth[j]);
      grad[i] = s;
      for (int j = 0; j < nv; j++) H[i][j] = d->M[i][j];
    }
    for (int r = 0; r < ne; r++)
      if (jar[r] < 0) {
        double w = d->efc_D[r];
        for (int i = 0; i < nv; i++) {
          double ji = d->efc_J[r][i];
          if (ji == 0.0) continue;
          grad[i] += w * jar[r] * ji;
          for (int j = 0; j < nv; j++) H[i][j] += w * ji * d->efc_J[r][j];
        }
      }
    double gn = 0;
    for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
    if (scale * sqrt(gn) < tol) break;
    if (chol_factor(H, nv)) { d->status |= MZ_STATUS_BAD_STATE; break; }
    for (int i = 0; i < nv; i++) search[i] = -grad[i];
    chol_solve(H, nv, search);
    /* exact line search on the piecewise-quadratic phi(alpha) */
    double p1 = 0, p2 = 0; /* phi'(a) = p1 + p2*a + sum_active D (jar + a jv) jv */
    for (int i = 0; i < nv; i++) {
      double s = 0, g0 = 0;
      for (int j = 0; j < nv; j++) { s += d->M[i][j] * search[j]; g0 += d->M[i][j] * (qacc[j] - d->qacc_smooth[j]); }
      Ms[i] = s;
      p1 += search[i] * g0;
      p2 += search[i] * s;
    }
    lsbp bp[MZO_MAX_EFC];
    int nb = 0;
    double q1 = p1, q2 = p2;
    for (int r = 0; r < ne; r++) {
      double v = 0;
      for (int i = 0; i < nv; i++) v += d->efc_J[r][i] * search[i];
      jv[r] = v;
      int act0 = jar[r] < 0 || (jar[r] == 0.0 && v < 0); /* active for alpha -> 0+ */
      if (act0) { q1 += d->efc_D[r] * jar[r] * v; q2 += d->efc_D[r] * v * v; }
      if (v != 0.0) {
        double a = -jar[r] / v;
        if (a > 0) { bp[nb].a = a; bp[nb].jar = jar[r]; bp[nb].jv = v; bp[nb].D = d->efc_D[r]; nb++; }
      }
    }
    qsort(bp, nb, sizeof(lsbp), cmp_bp);
    double alpha = 0;
    int found = 0;
    for (int k = 0; k <= nb; k++) {
      double hi = k < nb ? bp[k].a : INFINITY;
      double root = q2 > 0 ? -q1 / q2 : INFINITY;
      if (root <= hi) { alpha = root; found = 1; break; }
      if (k < nb) { /* crossing breakpoint k toggles that row */
        double sgn = (bp[k].jar < 0) ? -1.0 : 1.0; /* active -> inactive (-) or inactive -> active (+) */
        q1 += sgn * bp[k].D * bp[k].jar * bp[k].jv;
        q2 += sgn * bp[k].D * bp[k].jv * bp[k].jv;
      }
    }
    if (!found || !(alpha > 0) || !isfinite(alpha)) break;
    for (int i = 0; i < nv; i++) qacc[i] += alpha * search[i];
    for (int r = 0; r < ne; r++) jar[r] += alpha * jv[r];
    double newcost = eval_cost(m, d, qacc, jar); /* recompute jar exactly to avoid drift */
    d->solver_iter = it + 1;
    double improvement = scale * (cost - newcost);
    cost = newcost;
    if (!(improvement > 0)) break; /* round-off floor reached */
    if (it == maxiter - 1) d->status |= MZ_STATUS_SOLVER_MAXITER;
  }
  memcpy(d->qacc, qacc, sizeof(double) * nv);
  for (int r = 0; r < ne; r++) d->efc_force[r] = jar[r] < 0 ? -d->efc_D[r] * jar[r] : 0.0;
}

/* ------------------------------------------------------------------ forward dynamics */
void mzo_forward(const mz_model* m, mzo_data* d, const double* ctrl) {
  kinematics(m, d);
  com_and_crb(m, d);
  collision(m, d);
  make_constraints(m, d);
  velocity_and_forces(m, d, ctrl);
  int nv = m->nv;
  double L[ND][ND];
  for (int i = 0; i < nv; i++) {
    for (int j = 0; j < nv; j++) L[i][j] = d->M[i][j];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  if (chol_factor(L, nv)) d->status |= MZ_STATUS_BAD_STATE;
  chol_solve(L, nv, d->qacc_smooth);
  solve_constraints(m, d, d->solver_tol > 0 ? d->solver_tol : 1e-10, d->solver_maxiter > 0 ? d->solver_maxiter : 100);
}

static void integrate_pos(const mz_model* m, double* qpos, const double* vel, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == MZ_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * vel[da + k];
      double w[3] = {vel[da + 3], vel[da + 4], vel[da + 5]};
      double n = norm3(w);
      if (n > MINVAL) {
        double ax[3] = {w[0] / n, w[1] / n, w[2] / n}, qr[4], qn[4];
        axis_angle_quat(qr, ax, h * n);
        quat_mul(qn, qpos + qa + 3, qr);
        memcpy(qpos + qa + 3, qn, sizeof(qn));
      }
      quat_normalize(qpos + qa + 3);
    } else if (m->jnt_type[j] == MZ_JNT_BALL) { /* mj_integratePos: q <- q * exp(h w / 2), w in the child frame */
      double w[3] = {vel[da], vel[da + 1], vel[da + 2]};
      double n = norm3(w);
      if (n > MINVAL) {
        double ax[3] = {w[0] / n, w[1] / n, w[2] / n}, qr[4], qn[4];
        axis_angle_quat(qr, ax, h * n);
        quat_mul(qn, qpos + qa, qr);
        memcpy(qpos + qa, qn, sizeof(qn));
      }
      quat_normalize(qpos + qa);
    } else {
      qpos[qa] += h * vel[da];
    }
  }
}

static int bad(const double* x, int n) {
  for (int i = 0; i < n; i++)
    if (!isfinite(x[i]) || fabs(x[i]) > 1e10) return 1;
  return 0;
}

/* one mj_step: RK4 (every reference asset) or MuJoCo's default Euler (user robots) */
void mzo_mj_step(const mz_model* m, mzo_data* d, const double* ctrl) {
  int nq = m->nq, nv = m->nv;
  double h = m->timestep;
  static const double A[9] = {0.5, 0, 0, 0, 0.5, 0, 0, 0, 1.0}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  double X0q[MZ_MAX_Q], Xv[4][ND], F[4][ND], dv[ND], df[ND];
  if (bad(d->qpos, nq) || bad(d->qvel, nv)) d->status |= MZ_STATUS_BAD_STATE;
  mzo_forward(m, d, ctrl);
  if (bad(d->qacc, nv)) d->status |= MZ_STATUS_BAD_STATE;
  if (!m->integrator_rk4) {
    /* MuJoCo's default integrator (engine_forward.c mj_EulerSkip; >= 2.1.2 semantics [ASSUME-1]): semi-implicit Euler, implicit in
     * the joint damping — (M + h diag(damping)) qacc' = qfrc_smooth + qfrc_constraint = M qacc, then qvel += h qacc', qpos integrates
     * the NEW velocity; without damping qacc' = qacc.  qacc_warmstart keeps the solver's qacc (mj_advance). */
    double qa[ND], L[ND][ND];
    int damped = 0;
    for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0.0) damped = 1;
    for (int i = 0; i < nv; i++) qa[i] = d->qacc[i];
    if (damped) {
      for (int i = 0; i < nv; i++) {
        double t = 0.0;
        for (int j = 0; j < nv; j++) { t += d->M[i][j] * d->qacc[j]; L[i][j] = d->M[i][j]; }
        qa[i] = t;
        L[i][i] += h * m->dof_damping[i];
      }
      if (chol_factor(L, nv)) d->status |= MZ_STATUS_BAD_STATE;
      chol_solve(L, nv, qa);
    }
    for (int k = 0; k < nv; k++) d->qvel[k] += h * qa[k];
    integrate_pos(m, d->qpos, d->qvel, h);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
    d->time += h;
    return;
  }
  memcpy(X0q, d->qpos, sizeof(double) * nq);
  memcpy(Xv[0], d->qvel, sizeof(double) * nv);
  memcpy(F[0], d->qacc, sizeof(double) * nv);
  for (int i = 1; i < 4; i++) {
    for (int k = 0; k < nv; k++) {
      dv[k] = df[k] = 0;
      for (int j = 0; j < i; j++) { dv[k] += A[(i - 1) * 3 + j] * Xv[j][k]; df[k] += A[(i - 1) * 3 + j] * F[j][k]; }
    }
    memcpy(d->qpos, X0q, sizeof(double) * nq);
    integrate_pos(m, d->qpos, dv, h);
    for (int k = 0; k < nv; k++) { Xv[i][k] = Xv[0][k] + h * df[k]; d->qvel[k] = Xv[i][k]; }
    mzo_forward(m, d, ctrl);
    memcpy(F[i], d->qacc, sizeof(double) * nv);
  }
  for (int k = 0; k < nv; k++) {
    dv[k] = df[k] = 0;
    for (int j = 0; j < 4; j++) { dv[k] += B[j] * Xv[j][k]; df[k] += B[j] * F[j][k]; }
  }
  memcpy(d->qpos, X0q, sizeof(double) * nq);
  for (int k = 0; k < nv; k++) d->qvel[k] = Xv[0][k] + h * df[k];
  integrate_pos(m, d->qpos, dv, h);
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv); /* last stage's qacc, as mj_advance */
  d->time += h;
}

void mzo_data_init(const mz_model* m, mzo_data* d) {
  memset(d, 0, sizeof(*d));
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
}

/* kinetic + potential energy (for invariants tests) */
double mzo_energy(const mz_model* m, mzo_data* d) {
  kinematics(m, d);
  com_and_crb(m, d);
  double e = 0;
  for (int i = 0; i < m->nv; i++)
    for (int j = 0; j < m->nv; j++) e += 0.5 * d->qvel[i] * d->M[i][j] * d->qvel[j];
  for (int b = 1; b < m->nbody; b++) e -= m->body_mass[b] * dot3(m->gravity, d->xipos[b]);
  return e;
}
