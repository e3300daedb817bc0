// swimmer_dyn.h — the Swimmer robot's (and the Reacher's) MazeEnv.step, one environment per group of 4 (8) lanes (fp64).
//
// The Reacher (mujoco_maze/reacher.py:15-80, assets/reacher.xml — whose <mujoco model="swimmer">) is the same
// planar chain with TWO links and one motor, same medium, same step/reward code: the functions below are
// templated on the link count NL (3 = Swimmer, 2 = Reacher); dofs = NL + 2, inner hinges = motors = NL - 1.
//
// Replaces SwimmerEnv.step (mujoco_maze/swimmer.py:37-53: `do_simulation(action, 4)` = 4 x mj_step with
// RK4, h = 0.01; forward reward = |xy velocity|, ctrl cost 1e-4 |a|^2) and the MazeEnv.step bookkeeping
// around it.  Model (assets/swimmer.xml): a planar 3-link chain — torso with slide-x, slide-y and a hinge,
// two more hinged capsules (limits +-100 deg, motors with gear 150) — moving in a viscous medium
// (option density 4000, viscosity 0.1) with `collision="predefined"`, i.e. NO contacts (the maze walls
// do not stop a swimmer in the reference either).
//
// Dynamics per forward evaluation (16 per env.step):
//   * planar kinematics of the three link centres, their 2 x 5 Jacobians;
//   * M = sum_b m_b Jp^T Jp + Izz_b Jw^T Jw + armature (5 x 5 dense), bias = sum_b m_b Jp^T (centripetal accel);
//   * MuJoCo's inertia-box fluid forces per link (viscous + quadratic drag, in the link frame at its centre)
//     [ASSUME-9], motor torques with ctrl clamp;
//   * joint-limit rows on the two inner hinges as soft constraints, dense 5 x 5 Newton with exact line search;
//   * RK4 (all coordinates are scalars: no manifold update needed).
#pragma once
#include <math.h>
#include <stdint.h>

#include "ant_model.h"

// MZ_SW_MAXL: longest planar chain the kernels are instantiated for (user MJCF: mjcf.py accepts 2..6 links)
#define MZ_SW_MAXL 6

struct SwimmerDev {
  double h, density, viscosity, inv_scale;
  double gear[MZ_SW_MAXL - 1], ctrl_lo[MZ_SW_MAXL - 1], ctrl_hi[MZ_SW_MAXL - 1], armature[MZ_SW_MAXL + 2];  // per motor / per dof
  int frame_skip, reset_kind;
  double mass[MZ_SW_MAXL], izz[MZ_SW_MAXL], box[MZ_SW_MAXL][3];  // per link: mass, inertia about z through the centre, inertia-box sizes
  double com[MZ_SW_MAXL];             // link centre along the link's local x axis
  double off[MZ_SW_MAXL];             // origin of link b in its parent's frame, along the parent's local x
  double lim_lo[MZ_SW_MAXL - 1], lim_hi[MZ_SW_MAXL - 1], lim_K, lim_B, lim_solimp[5], dofw[MZ_SW_MAXL - 1];
  double qpos0[MZ_SW_MAXL + 2];
  int nlink;                          // 3 Swimmer, 2 Reacher; 4..6: user-supplied chains
  double mtot[2], inv_l[2];           // M00 / M11 = total mass + armature of the slide, and 1 / sqrt of them (sw_factor_slides)
  // movable blocks of the maze: `collision="predefined"` gives the swimmer no contact pairs at all, so a block is a free,
  // force-free slide-x / slide-y body — it only drifts with whatever velocity it is given and shows up in the observation
  int nblock, observe_blocks;
  double block_pos0[1][3], block_mass, block_box[3];  // inertia-box sizes: the medium drags a moving block too
  // slide dofs of the block: 2 (x, y: Push mazes) or 2 / 3 with a z slide (y, z: Fall; x, y, z: MultiFall), the latter LIMITED
  // (maze_env.py:607-648) and pulled by gravity
  int nbdof, bd_axis[3], bd_limited[3];
  double bd_lo[3], bd_hi[3], bd_margin, bd_K, bd_B, bd_solimp[5], gz;
  TaskDev task;
};

static inline int swimmer_dev_from_model(SwimmerDev* p, const mz_model* m, char* err, int errlen) {
  memset(p, 0, sizeof(*p));
  const int nbk = m->nblock, nl = m->nbody - 1 - nbk;
  if (nbk < 0 || nbk > 1) return ant_fail(err, errlen, "swimmer kernel: at most one movable block");
  const int nbd0 = nbk ? m->body_jntnum[m->block_bodyid[0]] : 0;
  bool ok = m->robot == MZ_ROBOT_SWIMMER && nl >= 2 && nl <= MZ_SW_MAXL && (nl <= 3 || nbk == 0) && m->nv == nl + 2 + nbd0 && m->nq == nl + 2 + nbd0 && m->nu == nl - 1 &&
            m->collision_predefined && m->jnt_type[0] == MZ_JNT_SLIDE && m->jnt_type[1] == MZ_JNT_SLIDE;
  for (int j = 2; ok && j < nl + 2; j++) ok = m->jnt_type[j] == MZ_JNT_HINGE;
  for (int a = 0; ok && a < nl - 1; a++) ok = m->act_dofid[a] == 3 + a;
  if (!ok) return ant_fail(err, errlen, "swimmer kernel: model is not a planar chain of 2..6 links on two slides (movable blocks: 2- and 3-link chains only)");
  p->nlink = nl; p->nblock = nbk; p->observe_blocks = m->observe_blocks;
  p->gz = m->gravity[2];
  const int nbd = nbk ? m->body_jntnum[m->block_bodyid[0]] : 0;
  for (int k = 0; k < nbk; k++) {
    int b = m->block_bodyid[k], j0 = m->body_jntadr[b];
    if (nbd < 2 || nbd > 3 || m->body_dofadr[b] != nl + 2) return ant_fail(err, errlen, "swimmer kernel: movable block needs 2 or 3 slide joints right after the robot");
    p->nbdof = nbd;
    for (int a = 0; a < nbd; a++) {
      int j = j0 + a, ax = -1;
      for (int c = 0; c < 3; c++) if (fabs(m->jnt_axis[j][c] - 1.0) < 1e-12) ax = c;
      if (m->jnt_type[j] != MZ_JNT_SLIDE || ax < 0 || (a > 0 && ax <= p->bd_axis[a - 1])) return ant_fail(err, errlen, "swimmer kernel: block joints must be slides along increasing coordinate axes");
      p->bd_axis[a] = ax; p->bd_limited[a] = m->jnt_limited[j]; p->bd_lo[a] = m->jnt_range[j][0]; p->bd_hi[a] = m->jnt_range[j][1];
      if (m->dof_armature[m->jnt_dofadr[j]] != 0.0 || m->dof_damping[m->jnt_dofadr[j]] != 0.0) return ant_fail(err, errlen, "swimmer kernel: block slides must be free of armature and damping");
    }
    {  // limit rows share the joint defaults (maze_env.py:607-648 sets only margin)
      double tc = fmax(m->jnt_solref[j0][0], 2.0 * m->timestep), dr = m->jnt_solref[j0][1], dmax = m->jnt_solimp[j0][1];
      p->bd_K = 1.0 / (dmax * dmax * tc * tc * dr * dr); p->bd_B = 2.0 / (dmax * tc); p->bd_margin = m->jnt_margin[j0];
      for (int q = 0; q < 5; q++) p->bd_solimp[q] = m->jnt_solimp[j0][q];
    }
    for (int q = 0; q < 3; q++) p->block_pos0[k][q] = m->body_pos[b][q];
    const double* I = m->body_inertia[b];
    p->block_mass = m->body_mass[b];
    p->block_box[0] = sqrt(fmax(1e-15, I[1] + I[2] - I[0]) / p->block_mass * 6.0);
    p->block_box[1] = sqrt(fmax(1e-15, I[0] + I[2] - I[1]) / p->block_mass * 6.0);
    p->block_box[2] = sqrt(fmax(1e-15, I[0] + I[1] - I[2]) / p->block_mass * 6.0);
  }
  p->h = m->timestep; p->frame_skip = m->frame_skip; p->reset_kind = m->reset_qvel_kind;
  for (int a = 0; a < nl - 1; a++) { p->gear[a] = m->act_gear[a]; p->ctrl_lo[a] = m->act_ctrlrange[a][0]; p->ctrl_hi[a] = m->act_ctrlrange[a][1]; }
  for (int k = 0; k < nl + 2; k++) {
    p->armature[k] = m->dof_armature[k];
    if (m->dof_damping[k] != 0.0) return ant_fail(err, errlen, "swimmer kernel: joint damping is not modelled (the assets have none)");
  }
  p->density = m->density; p->viscosity = m->viscosity;
  p->inv_scale = 1.0 / (m->meaninertia * m->nv);  // MuJoCo scales the solver tolerance with the whole model
  for (int b = 0; b < nl; b++) {
    const double* I = m->body_inertia[b + 1];
    if (fabs(m->body_ipos[b + 1][1]) > 1e-12 || fabs(I[3]) + fabs(I[4]) + fabs(I[5]) > 1e-12)
      return ant_fail(err, errlen, "swimmer kernel: links must lie on their local x axis");
    p->mass[b] = m->body_mass[b + 1]; p->izz[b] = I[2]; p->com[b] = m->body_ipos[b + 1][0];
    p->off[b] = b == 0 ? 0.0 : m->body_pos[b + 1][0];
    p->box[b][0] = sqrt(fmax(1e-15, I[1] + I[2] - I[0]) / p->mass[b] * 6.0);
    p->box[b][1] = sqrt(fmax(1e-15, I[0] + I[2] - I[1]) / p->mass[b] * 6.0);
    p->box[b][2] = sqrt(fmax(1e-15, I[0] + I[1] - I[2]) / p->mass[b] * 6.0);
  }
  for (int k = 0; k < nl - 1; k++) {
    int j = 3 + k;
    if (!m->jnt_limited[j]) return ant_fail(err, errlen, "swimmer kernel: inner hinges must be limited");
    p->lim_lo[k] = m->jnt_range[j][0]; p->lim_hi[k] = m->jnt_range[j][1]; p->dofw[k] = m->dof_invweight0[j];
  }
  double tc = fmax(m->jnt_solref[3][0], 2.0 * m->timestep), dr = m->jnt_solref[3][1], dmax = m->jnt_solimp[3][1];
  p->lim_K = 1.0 / (dmax * dmax * tc * tc * dr * dr);
  p->lim_B = 2.0 / (dmax * tc);
  for (int k = 0; k < 5; k++) p->lim_solimp[k] = m->jnt_solimp[3][k];
  for (int k = 0; k < nl + 2; k++) p->qpos0[k] = m->qpos0[k];
  for (int a = 0; a < 2; a++) {
    double t = p->armature[a];
    for (int b = 0; b < nl; b++) t += p->mass[b];
    p->mtot[a] = t; p->inv_l[a] = 1.0 / sqrt(t);
  }
  task_dev_from_model(&p->task, m);
  return MZ_OK;
}

#if defined(__HIPCC__)
#define MZS_HD __host__ __device__ __forceinline__
#else
#define MZS_HD inline
#endif

MZS_HD double sw_impedance(const double* si, double x) {
  double d0 = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
  if (d0 == dmax || width <= 1e-15) return 0.5 * (d0 + dmax);
  double xn = x / width;
  if (xn >= 1.0) return dmax;
  if (xn <= 0.0) return d0;
  double y;
  if (power <= 1.0 + 1e-12) y = xn;
  else if (power == 2.0) y = xn <= mid ? xn * xn / mid : 1.0 - (1.0 - xn) * (1.0 - xn) / (1.0 - mid);  // MuJoCo's default power: no pow()
  else if (xn <= mid) y = pow(xn, power) / pow(mid, power - 1.0);
  else y = 1.0 - pow(1.0 - xn, power) / pow(1.0 - mid, power - 1.0);
  return d0 + y * (dmax - d0);
}

// dense symmetric positive definite solve A x = b (n = NV <= 5), Cholesky in registers
// (one reciprocal square root per column and no division: the columns are scaled by 1 / L_jj, which is also all the two
// substitutions need — on the device v_rsq_f64 + refinement instead of five square roots and ten reciprocals in float64)
MZS_HD double sw_rsqrt(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rsqrt(d);
#else
  return 1.0 / sqrt(d);
#endif
}
template <int NV>
MZS_HD void sw_subst(const double L[NV][NV], const double* inv, const double* b, double* x) {
  double y[NV];
  for (int i = 0; i < NV; i++) { double t = b[i]; for (int k = 0; k < i; k++) t -= L[i][k] * y[k]; y[i] = t * inv[i]; }
  for (int i = NV - 1; i >= 0; i--) { double t = y[i]; for (int k = i + 1; k < NV; k++) t -= L[k][i] * x[k]; x[i] = t * inv[i]; }
}
// columns J0 .. NV-1 of the factor (the columns before them are taken as they stand in L / inv)
template <int NV, int J0>
MZS_HD void sw_factor_from(const double A[NV][NV], double L[NV][NV], double* inv) {
  for (int j = J0; j < NV; j++) {
    double d = A[j][j];
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
    inv[j] = sw_rsqrt(fmax(d, 1e-300));
    for (int i = j + 1; i < NV; i++) {
      double t = A[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t * inv[j];
    }
  }
}
template <int NV>
MZS_HD void sw_solve(const double A[NV][NV], const double* b, double* x) {
  double L[NV][NV], inv[NV];
  sw_factor_from<NV, 0>(A, L, inv);
  sw_subst<NV>(L, inv, b, x);
}
// The chain's mass matrix has a fixed corner: the two slides carry the whole mass and do not couple (M00 = M11 = total mass +
// armature, M10 = 0), so the first two columns of the factor need no square root — inv0 / inv1 come from the host — and the
// second column no elimination.  Used by the lane-group path; the generic sw_solve stays the one-lane path's.
template <int NV>
MZS_HD void sw_factor_slides(const double A[NV][NV], double inv0, double inv1, double L[NV][NV], double* inv) {
  inv[0] = inv0; inv[1] = inv1;
  L[1][0] = 0.0;
  for (int i = 2; i < NV; i++) { L[i][0] = A[i][0] * inv0; L[i][1] = A[i][1] * inv1; }
  sw_factor_from<NV, 2>(A, L, inv);
}

// Execution context of the chain's forward dynamics.  On the device a group of G adjacent lanes (4 for chains of up to four
// links, 8 beyond) advances ONE environment: lane b of the group owns link b — its orientation's sine / cosine, its centre's
// Jacobian, its fluid wrench, its share of M and of the force vector — and the shares meet in group sums (DPP moves on the two
// halves of a double, mz_device.h); the 5 x 5 solve and the joint-limit Newton then run redundantly on every lane of the group,
// as does the RK4 bookkeeping (state in registers: nothing is handed through memory).  The one-lane context (host emulation,
// tests/emu) walks all links itself.
struct SwimmerOneLane {
  static constexpr int nlanes = 1;
  MZS_HD int lane0() const { return 0; }
  MZS_HD double gsum(double x) const { return x; }
  MZS_HD double from_lane(double x, int) const { return x; }
  MZS_HD void stamp(int) const {}  // phase timers of experiment builds (planar_kernels.hip, MZ_EXP_SWPROF)
};

// forward dynamics: qacc from (q, v, motor torques tau[NL - 1]); returns status bits
template <int NL, class C>
MZS_HD int swimmer_forward(const C& cx, const SwimmerDev& P, const double* q, const double* v, const double* tau, double* qacc) {
  constexpr int NV = NL + 2, NH = NL - 1;
  cx.stamp(5);
  // link orientation angles and rates
  double phi[NL], om[NL], c[NL], s[NL];
  phi[0] = q[2]; om[0] = v[2];
  for (int b = 1; b < NL; b++) { phi[b] = phi[b - 1] + q[2 + b]; om[b] = om[b - 1] + v[2 + b]; }
  double cm = 1.0, sm = 0.0;  // lane-group path: cosine / sine of this lane's own link
  if constexpr (C::nlanes == 1) {
    for (int b = 0; b < NL; b++) { c[b] = cos(phi[b]); s[b] = sin(phi[b]); }
  } else {  // each lane evaluates the sine / cosine of its own link's angle; the others come from their lanes
    const int me = cx.lane0() < NL ? cx.lane0() : NL - 1;
    double pm = phi[0];
#pragma unroll
    for (int b = 1; b < NL; b++) pm = me == b ? phi[b] : pm;
    sincos(pm, &sm, &cm);  // one argument reduction for both
#pragma unroll
    for (int b = 0; b < NL; b++) { c[b] = cx.from_lane(cm, b); s[b] = cx.from_lane(sm, b); }
  }
  cx.stamp(0);
  double M[NV][NV], frc[NV], L[NV][NV], inv[NV];
  if constexpr (C::nlanes > 1) {
    // Lane-group path: ONE instruction stream for every link — lane b holds link b's constants and chain length in registers
    // (selected once; a per-link `if (b != lane) continue` would make the wave walk all NL bodies one after the other with a
    // quarter of its lanes enabled each time).  p_b = p0 + sum_{k<b} off[k+1] e(phi_k) + com[b] e(phi_b): segment k of lane b
    // has length off[k+1] (k < b), com[b] (k == b) or 0 (k > b), so the loop below is the same for all lanes.  Lanes beyond
    // the chain's end carry a massless, sizeless copy of the last link: their shares are zeros.
    const int ln = cx.lane0(), lb = ln < NL ? ln : NL - 1;
    const bool act = ln < NL;
    double m = P.mass[0], izz = P.izz[0], com = P.com[0], bx0 = P.box[0][0], bx1 = P.box[0][1], bx2 = P.box[0][2], w = om[0];
#pragma unroll
    for (int b = 1; b < NL; b++) {
      const bool me = lb == b;
      m = me ? P.mass[b] : m; izz = me ? P.izz[b] : izz; com = me ? P.com[b] : com; w = me ? om[b] : w;
      bx0 = me ? P.box[b][0] : bx0; bx1 = me ? P.box[b][1] : bx1; bx2 = me ? P.box[b][2] : bx2;
    }
    if (!act) { m = 0.0; izz = 0.0; bx0 = 0.0; bx1 = 0.0; bx2 = 0.0; }
    double Jx[NL], Jy[NL], ax = 0.0, ay = 0.0;  // columns of the hinge dofs 2 .. NV-1 (the slides' columns are unit vectors)
#pragma unroll
    for (int j = 0; j < NL; j++) { Jx[j] = 0.0; Jy[j] = 0.0; }
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const double len = k < lb ? P.off[k + 1 < NL ? k + 1 : 0] : (k == lb ? com : 0.0);
      const double ex = c[k] * len, ey = s[k] * len;
#pragma unroll
      for (int j = 0; j <= k; j++) { Jx[j] -= ey; Jy[j] += ex; }
      ax -= om[k] * om[k] * ex; ay -= om[k] * om[k] * ey;  // centripetal acceleration at qacc = 0
    }
    double vx = v[0], vy = v[1];
#pragma unroll
    for (int j = 0; j < NL; j++) { vx += Jx[j] * v[2 + j]; vy += Jy[j] * v[2 + j]; }
    const double co = cm, so = sm;  // the fluid forces act in the link frame at its centre (MuJoCo inertia-box model)
    const double lvx = co * vx + so * vy, lvy = -so * vx + co * vy;
    double fx = 0.0, fy = 0.0, tz = 0.0;
    if (P.viscosity > 0.0) {
      const double diam = (bx0 + bx1 + bx2) / 3.0;
      const double sl = -3.0 * P.viscosity * 3.141592653589793 * diam, sa = -P.viscosity * 3.141592653589793 * diam * diam * diam;
      fx += sl * lvx; fy += sl * lvy; tz += sa * w;
    }
    if (P.density > 0.0) {
      fx -= 0.5 * P.density * bx1 * bx2 * fabs(lvx) * lvx;
      fy -= 0.5 * P.density * bx0 * bx2 * fabs(lvy) * lvy;
      tz -= P.density * bx2 * (bx0 * bx0 * bx0 * bx0 + bx1 * bx1 * bx1 * bx1) * fabs(w) * w / 64.0;
    }
    const double Gx = (co * fx - so * fy) - m * ax, Gy = (so * fx + co * fy) - m * ay;
    cx.stamp(1);
    // shares -> group sums; the slides' corner of M is constant (SwimmerDev::mtot)
    frc[0] = cx.gsum(Gx); frc[1] = cx.gsum(Gy);
    M[0][0] = P.mtot[0]; M[1][1] = P.mtot[1]; M[1][0] = 0.0; M[0][1] = 0.0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const bool on = i <= lb;  // hinge i turns this lane's link
      frc[2 + i] = cx.gsum(Jx[i] * Gx + Jy[i] * Gy + (on ? tz : 0.0));
      const double t0 = cx.gsum(m * Jx[i]), t1 = cx.gsum(m * Jy[i]);
      M[2 + i][0] = t0; M[0][2 + i] = t0; M[2 + i][1] = t1; M[1][2 + i] = t1;
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double t = cx.gsum(m * (Jx[i] * Jx[j] + Jy[i] * Jy[j]) + (on ? izz : 0.0));
        if (i == j) t += P.armature[2 + i];
        M[2 + i][2 + j] = t; M[2 + j][2 + i] = t;
      }
    }
    for (int k = 0; k < NH; k++) frc[3 + k] += tau[k];
    cx.stamp(2);
    if constexpr (NL > 3) sw_factor_slides<NV>(M, P.inv_l[0], P.inv_l[1], L, inv);
  } else {
  // position Jacobians of the link centres: p_b = p0 + sum_{k<b} off[k+1] e(phi_k) + com[b] e(phi_b)
  for (int i = 0; i < NV; i++) {
    frc[i] = 0.0;
    for (int j = 0; j < NV; j++) M[i][j] = 0.0;
  }
  for (int b = 0; b < NL; b++) {
    double Jx[NV], Jy[NV], ax = 0.0, ay = 0.0;
    for (int k = 0; k < NV; k++) { Jx[k] = 0.0; Jy[k] = 0.0; }
    Jx[0] = 1.0; Jy[1] = 1.0;
    for (int k = 0; k <= b; k++) {
      const double len = k < b ? P.off[k + 1] : P.com[b];  // segment carried by the frame of link k
      const double ex = c[k] * len, ey = s[k] * len;
      // d/d(phi_k) of the segment = (-ey, ex); phi_k depends on the hinge dofs 2..2+k
      for (int d = 2; d <= 2 + k; d++) { Jx[d] += -ey; Jy[d] += ex; }
      ax += -om[k] * om[k] * ex; ay += -om[k] * om[k] * ey;  // centripetal acceleration at qacc = 0
    }
    double vx = 0.0, vy = 0.0;
    for (int k = 0; k < NV; k++) { vx += Jx[k] * v[k]; vy += Jy[k] * v[k]; }
    const double m = P.mass[b];
    // fluid forces in the link frame at its centre (MuJoCo inertia-box model)
    const double lvx = c[b] * vx + s[b] * vy, lvy = -s[b] * vx + c[b] * vy, w = om[b];
    const double* bx = P.box[b];
    double fx = 0.0, fy = 0.0, tz = 0.0;
    if (P.viscosity > 0.0) {
      const double diam = (bx[0] + bx[1] + bx[2]) / 3.0;
      const double sl = -3.0 * P.viscosity * 3.141592653589793 * diam, sa = -P.viscosity * 3.141592653589793 * diam * diam * diam;
      fx += sl * lvx; fy += sl * lvy; tz += sa * w;
    }
    if (P.density > 0.0) {
      fx -= 0.5 * P.density * bx[1] * bx[2] * fabs(lvx) * lvx;
      fy -= 0.5 * P.density * bx[0] * bx[2] * fabs(lvy) * lvy;
      tz -= P.density * bx[2] * (bx[0] * bx[0] * bx[0] * bx[0] + bx[1] * bx[1] * bx[1] * bx[1]) * fabs(w) * w / 64.0;
    }
    const double Fx = c[b] * fx - s[b] * fy, Fy = s[b] * fx + c[b] * fy;
    for (int i = 0; i < NV; i++) {
      const double jw_i = (i >= 2 && i <= 2 + b) ? 1.0 : 0.0;
      frc[i] += Jx[i] * (Fx - m * ax) + Jy[i] * (Fy - m * ay) + jw_i * tz;
      for (int j = 0; j <= i; j++) {
        const double jw_j = (j >= 2 && j <= 2 + b) ? 1.0 : 0.0;
        M[i][j] += m * (Jx[i] * Jx[j] + Jy[i] * Jy[j]) + P.izz[b] * jw_i * jw_j;
      }
    }
  }
  for (int i = 0; i < NV; i++)
    for (int j = 0; j <= i; j++) { const double t = M[i][j] + (i == j ? P.armature[i] : 0.0); M[i][j] = t; M[j][i] = t; }
  for (int k = 0; k < NH; k++) frc[3 + k] += tau[k];
  if constexpr (NL > 3) sw_factor_from<NV, 0>(M, L, inv);
  }
  double qas[NV];
  // Swimmer / Reacher (two or three links): the slides are eliminated by hand (their block of M is diagonal) and the NL x NL
  // Schur complement of the hinges is inverted by cofactors — one reciprocal, independent products — instead of a factorisation
  // whose columns wait for one another; Si = (M^-1) restricted to the hinges is also what the limit rows below need.
  double Si[NL <= 3 ? NL : 1][NL <= 3 ? NL : 1], i0 = 0.0, i1 = 0.0;
  if constexpr (NL <= 3) {
    if constexpr (C::nlanes > 1) { i0 = P.inv_l[0] * P.inv_l[0]; i1 = P.inv_l[1] * P.inv_l[1]; }
    else { i0 = 1.0 / M[0][0]; i1 = 1.0 / M[1][1]; }
    double S[NL][NL], g[NL];
    for (int i = 0; i < NL; i++) {
      g[i] = frc[2 + i] - M[2 + i][0] * (frc[0] * i0) - M[2 + i][1] * (frc[1] * i1);
      for (int j = 0; j <= i; j++) S[i][j] = M[2 + i][2 + j] - (M[2 + i][0] * i0) * M[2 + j][0] - (M[2 + i][1] * i1) * M[2 + j][1];
    }
    if constexpr (NL == 3) {
      const double c00 = S[1][1] * S[2][2] - S[2][1] * S[2][1], c10 = S[2][1] * S[2][0] - S[1][0] * S[2][2], c20 = S[1][0] * S[2][1] - S[1][1] * S[2][0];
      const double c11 = S[0][0] * S[2][2] - S[2][0] * S[2][0], c21 = S[1][0] * S[2][0] - S[0][0] * S[2][1], c22 = S[0][0] * S[1][1] - S[1][0] * S[1][0];
      const double id = 1.0 / (S[0][0] * c00 + S[1][0] * c10 + S[2][0] * c20);
      Si[0][0] = c00 * id; Si[1][1] = c11 * id; Si[2][2] = c22 * id;
      Si[1][0] = Si[0][1] = c10 * id; Si[2][0] = Si[0][2] = c20 * id; Si[2][1] = Si[1][2] = c21 * id;
    } else {
      const double id = 1.0 / (S[0][0] * S[1][1] - S[1][0] * S[1][0]);
      Si[0][0] = S[1][1] * id; Si[1][1] = S[0][0] * id; Si[1][0] = Si[0][1] = -S[1][0] * id;
    }
    double s0 = frc[0], s1 = frc[1];
    for (int i = 0; i < NL; i++) {
      double t = 0.0;
      for (int j = 0; j < NL; j++) t += Si[i][j] * g[j];
      qas[2 + i] = t; s0 -= M[2 + i][0] * t; s1 -= M[2 + i][1] * t;
    }
    qas[0] = s0 * i0; qas[1] = s1 * i1;
  } else {
    sw_subst<NV>(L, inv, frc, qas);
  }
  for (int i = 0; i < NV; i++) qacc[i] = qas[i];
  cx.stamp(3);
  // joint limits on the inner hinges
  double sg[NH], D[NH], aref[NH];
  for (int k = 0; k < NH; k++) { sg[k] = 0.0; D[k] = 0.0; aref[k] = 0.0; }
  bool any = false;
  for (int k = 0; k < NH; k++) {
    double qq = q[3 + k], pos = 0.0;
    if (qq - P.lim_lo[k] < 0.0) { sg[k] = 1.0; pos = qq - P.lim_lo[k]; }
    else if (P.lim_hi[k] - qq < 0.0) { sg[k] = -1.0; pos = P.lim_hi[k] - qq; }
    if (sg[k] != 0.0) {
      double imp = sw_impedance(P.lim_solimp, fabs(pos));
      D[k] = 1.0 / fmax(1e-15, (1.0 - imp) / imp * P.dofw[k]);
      aref[k] = -P.lim_B * (sg[k] * v[3 + k]) - P.lim_K * imp * pos;
      any = true;
    }
  }
  cx.stamp(4);
  if (!any) return 0;
  if constexpr (NH <= 2) {
    // One or two limit rows, each on a single dof (J_k = sg_k e_{3+k}): the minimiser of
    //   1/2 (a - a_s)^T M (a - a_s) + sum_k D_k / 2 min(0, sg_k a_{3+k} - aref_k)^2
    // is a = a_s + M^-1 sum_k sg_k lam_k e_{3+k} with lam_k = D_k max(0, -r_k) >= 0, r = r0 + G lam, r0_k = sg_k a_s[3+k] - aref_k,
    // G = S (M^-1)_hinge S: a piecewise linear system in at most two unknowns, solved by trying its active sets (both rows, one
    // row, none — strict convexity makes exactly one of them consistent).  Same optimum as the Newton iteration below (which the
    // longer chains keep, and which the oracle runs to 1e-10), without its factorisations and line searches.
    double X0[NV], r0[NH], Rk[NH];
    bool val[NH];
    for (int k = 0; k < NH; k++) { val[k] = sg[k] != 0.0; r0[k] = val[k] ? sg[k] * qas[3 + k] - aref[k] : 0.0; Rk[k] = val[k] ? 1.0 / D[k] : 1.0; }
    // X_k = M^-1 e_{3+k}: hinge part = column 1 + k of Si, slide part from the eliminated rows
    auto unit = [&](int col, double* X) {
      double t0 = 0.0, t1 = 0.0;
      for (int i = 0; i < NL; i++) { X[2 + i] = Si[i][col]; t0 -= M[2 + i][0] * Si[i][col]; t1 -= M[2 + i][1] * Si[i][col]; }
      X[0] = t0 * i0; X[1] = t1 * i1;
    };
    unit(1, X0);
    if constexpr (NH == 1) {
      const double lam = (val[0] && r0[0] < 0.0) ? -r0[0] / (X0[3] + Rk[0]) : 0.0;  // G00 = sg^2 (M^-1)_33
      for (int i = 0; i < NV; i++) qacc[i] = qas[i] + (sg[0] * lam) * X0[i];
    } else {
      double X1[NV];
      unit(2, X1);
      const double g00 = X0[3] + Rk[0], g11 = X1[4] + Rk[1], g01 = sg[0] * sg[1] * X1[3];
      const double idet = 1.0 / (g00 * g11 - g01 * g01);
      const double lb0 = (-r0[0] * g11 + r0[1] * g01) * idet, lb1 = (-r0[1] * g00 + r0[0] * g01) * idet;  // both rows active
      const double ls0 = -r0[0] / g00, ls1 = -r0[1] / g11;                                              // one row active
      const bool both = val[0] && val[1] && lb0 > 0.0 && lb1 > 0.0;
      const bool only0 = val[0] && r0[0] < 0.0 && (!val[1] || r0[1] + g01 * ls0 >= 0.0);
      const bool only1 = val[1] && r0[1] < 0.0 && (!val[0] || r0[0] + g01 * ls1 >= 0.0);
      const bool none = !(val[0] && r0[0] < 0.0) && !(val[1] && r0[1] < 0.0);
      double l0, l1;
      if (both) { l0 = lb0; l1 = lb1; }
      else if (only0) { l0 = ls0; l1 = 0.0; }
      else if (only1) { l0 = 0.0; l1 = ls1; }
      else if (none) { l0 = 0.0; l1 = 0.0; }
      else {  // on the boundary between two active sets (rounding): their solutions agree there
        l0 = val[0] ? fmax(0.0, val[1] ? lb0 : ls0) : 0.0;
        l1 = val[1] ? fmax(0.0, val[0] ? lb1 : ls1) : 0.0;
      }
      for (int i = 0; i < NV; i++) qacc[i] = qas[i] + (sg[0] * l0) * X0[i] + (sg[1] * l1) * X1[i];
    }
    cx.stamp(4);
    return 0;
  }
  int status = 0;
  for (int it = 0; it < 50; it++) {
    double grad[NV], H[NV][NV], jar[NH], act[NH];
    for (int i = 0; i < NV; i++) {
      grad[i] = 0.0;
      for (int j = 0; j < NV; j++) { grad[i] += M[i][j] * (qacc[j] - qas[j]); H[i][j] = M[i][j]; }
    }
    for (int k = 0; k < NH; k++) {
      jar[k] = sg[k] != 0.0 ? sg[k] * qacc[3 + k] - aref[k] : 0.0;
      act[k] = (sg[k] != 0.0 && jar[k] < 0.0) ? D[k] : 0.0;
      grad[3 + k] += sg[k] * act[k] * jar[k];
      H[3 + k][3 + k] += act[k];
    }
    double gn = 0.0;
    for (int i = 0; i < NV; i++) gn += grad[i] * grad[i];
    if (P.inv_scale * sqrt(gn) < 1e-10) break;
    if (it == 49) status |= MZ_STATUS_SOLVER_MAXITER;
    double sr[NV], ng[NV];
    for (int i = 0; i < NV; i++) ng[i] = -grad[i];
    sw_factor_from<NV, 3>(H, L, inv);  // H = M + diag on the inner hinges (dofs 3 .. NV-1): columns 0..2 of the factor are M's
    sw_subst<NV>(L, inv, ng, sr);
    double p1 = 0.0, p2 = 0.0;
    for (int i = 0; i < NV; i++) {
      double ms = 0.0, mx = 0.0;
      for (int j = 0; j < NV; j++) { ms += M[i][j] * sr[j]; mx += M[i][j] * (qacc[j] - qas[j]); }
      p1 += sr[i] * mx; p2 += sr[i] * ms;
    }
    double lo = 0.0, hi = -1.0, alpha = 1.0, prev_d2 = -1.0;
    for (int ls = 0; ls < 30; ls++) {
      double d1 = p1 + alpha * p2, d2 = p2;
      for (int k = 0; k < NH; k++)
        if (sg[k] != 0.0) {
          double jv = sg[k] * sr[3 + k], r = jar[k] + alpha * jv;
          if (r < 0.0) { d1 += D[k] * r * jv; d2 += D[k] * jv * jv; }
        }
      if (d2 == prev_d2) break;
      prev_d2 = d2;
      if (d1 < 0) lo = alpha; else hi = alpha;
      double next = alpha - d1 / d2;
      if (hi >= 0 && !(next > lo && next < hi)) next = 0.5 * (lo + hi);
      if (!(next > 0)) next = hi >= 0 ? 0.5 * (lo + hi) : 0.0;
      if (fabs(next - alpha) <= 1e-15 * fabs(next)) { alpha = next; break; }
      alpha = next;
    }
    for (int i = 0; i < NV; i++) qacc[i] += alpha * sr[i];
  }
  cx.stamp(4);
  return status;
}

// A movable block in the swimmer's world: no contacts (collision="predefined"), so each of its slide dofs evolves on its own:
//   m a = -(3 pi mu d v + rho/2 A |v| v)   the medium's drag on the box (inertia-box model, axis-aligned, no rotation)
//         + m g                            on a z slide (Fall / MultiFall blocks)
//         + f_limit                        soft joint-limit row of a LIMITED slide: one-row problem solved in closed form —
//                                          minimise m/2 (a - a0)^2 + D/2 min(0, J a - aref)^2  =>  a = (m a0 + D J aref) / (m + D)
// One env.step = frame_skip RK4 steps per dof.  (The drag on a 0.2 g / 1 g box is far stiffer than 1 / h: once the block
// moves the explicit integration diverges, in MuJoCo as here — the env is then flagged.)
MZS_HD double swimmer_block_acc(const SwimmerDev& P, int a, double q, double v) {
  const double* bx = P.block_box;
  const int ax = P.bd_axis[a];
  const double diam = (bx[0] + bx[1] + bx[2]) / 3.0, lin = 3.0 * P.viscosity * 3.141592653589793 * diam;
  const double area = ax == 0 ? bx[1] * bx[2] : (ax == 1 ? bx[0] * bx[2] : bx[0] * bx[1]);
  double a0 = -(lin * v + 0.5 * P.density * area * fabs(v) * v) / P.block_mass + (ax == 2 ? P.gz : 0.0);
  if (P.bd_limited[a]) {
    for (int side = -1; side <= 1; side += 2) {
      const double dist = side < 0 ? q - P.bd_lo[a] : P.bd_hi[a] - q;
      if (dist < P.bd_margin) {
        const double J = -(double)side, imp = sw_impedance(P.bd_solimp, fabs(dist - P.bd_margin));
        const double D = 1.0 / fmax(1e-15, (1.0 - imp) / imp * (1.0 / P.block_mass));  // dof_invweight0 of a free slide = 1 / m
        const double aref = -P.bd_B * (J * v) - P.bd_K * imp * (dist - P.bd_margin);
        if (J * a0 - aref < 0.0) a0 = (P.block_mass * a0 + D * J * aref) / (P.block_mass + D);
      }
    }
  }
  return a0;
}
MZS_HD void swimmer_block_step(const SwimmerDev& P, double* qb, double* vb) {
  for (int a = 0; a < P.nbdof; a++) {
    double q = qb[a], v = vb[a];
    for (int f = 0; f < P.frame_skip; f++) {
      double qs = q, vs = v, accv = 0.0, accf = 0.0;
      for (int st = 0; st < 4; st++) {
        double acc = swimmer_block_acc(P, a, qs, vs);
        double bw = (st == 0 || st == 3) ? 1.0 / 6 : 1.0 / 3, aw = st == 2 ? 1.0 : 0.5;
        accv += bw * vs; accf += bw * acc;
        double nq = q + P.h * (aw * vs), nv = v + P.h * (aw * acc);
        qs = nq; vs = nv;
      }
      q += P.h * accv; v += P.h * accf;
    }
    qb[a] = q; vb[a] = v;
  }
}

// One MazeEnv.step: q, v in/out (fp64 working copy); info4 = x, y, reward_forward, reward_ctrl
template <int NL, class C>
MZS_HD int swimmer_env_step(const C& cx, const SwimmerDev& P, double* q, double* v, const double* action, int t_in, double* inner_reward,
                            double* info4, int* t_out) {
  int status = 0;
  double x0 = q[0], y0 = q[1];
  constexpr int NV = NL + 2, NH = NL - 1;
  double tau[NH];
  for (int k = 0; k < NH; k++) tau[k] = P.gear[k] * fmin(fmax(action[k], P.ctrl_lo[k]), P.ctrl_hi[k]);
  for (int f = 0; f < P.frame_skip; f++) {
    const double h = P.h;
    double q0[NV], v0[NV], accv[NV], accf[NV], qs[NV], vs[NV], a[NV];
    for (int k = 0; k < NV; k++) { q0[k] = q[k]; v0[k] = v[k]; accv[k] = 0.0; accf[k] = 0.0; qs[k] = q[k]; vs[k] = v[k]; }
    for (int st = 0; st < 4; st++) {
      status |= swimmer_forward<NL>(cx, P, qs, vs, tau, a);
      double bw = (st == 0 || st == 3) ? 1.0 / 6 : 1.0 / 3, aw = st == 2 ? 1.0 : 0.5;
      for (int k = 0; k < NV; k++) {
        accv[k] += bw * vs[k]; accf[k] += bw * a[k];
        double nq = q0[k] + h * (aw * vs[k]), nv = v0[k] + h * (aw * a[k]);
        qs[k] = nq; vs[k] = nv;
      }
    }
    for (int k = 0; k < NV; k++) { q[k] = q0[k] + h * accv[k]; v[k] = v0[k] + h * accf[k]; }
  }
  double dt = P.h * P.frame_skip;
  double vx = (q[0] - x0) / dt, vy = (q[1] - y0) / dt;
  double fwd = sqrt(vx * vx + vy * vy), cc = 0.0;
  for (int k = 0; k < NH; k++) cc += action[k] * action[k];
  cc *= P.task.ctrl_w;
  *inner_reward = P.task.fwd_w * fwd - cc;
  info4[0] = q[0]; info4[1] = q[1]; info4[2] = fwd; info4[3] = -cc;
  *t_out = t_in + 1;
  return status;
}

// Observation row (swimmer.py:50-54 returns the WHOLE qpos and qvel — the slides of a movable block included — and
// MazeEnv._get_obs, maze_env.py:351-369, splices the block position in after the first three entries):
//   qpos[0:3] | block xyz (if observed) | qpos[3:NV] | qvel[0:NV] | t * 0.001        (SwimmerPush: 18 numbers)
// BD = slide dofs of the (single) movable block: 0 none, 2 or 3.
template <int NL, int BD>
MZS_HD void swimmer_obs_row(const SwimmerDev& P, const float* qf, const float* vf, int t, float* o) {
  constexpr int NV = NL + 2 + BD;
  const int nb3 = (BD && P.observe_blocks) ? 3 : 0;
  for (int k = 0; k < 3; k++) o[k] = qf[k];
  if (nb3) {
    double p[3] = {P.block_pos0[0][0], P.block_pos0[0][1], P.block_pos0[0][2]};
    for (int a = 0; a < BD; a++) p[P.bd_axis[a]] += (double)qf[NL + 2 + a];
    for (int c = 0; c < 3; c++) o[3 + c] = (float)p[c];
  }
  for (int k = 3; k < NV; k++) o[nb3 + k] = qf[k];
  for (int k = 0; k < NV; k++) o[nb3 + NV + k] = vf[k];
  o[nb3 + 2 * NV] = (float)t * 0.001f;
}

// One MazeEnv.step of a swimmer / reacher env with a BD-dof movable block on the fp32 state (qf, vf: NV = NL + 2 + BD entries,
// in and out): the chain's step, the block's own motion, the observation row o[2 NV + 4] and the inner reward.
// Returns status bits.  Shared by swimmer_step_kernel and the CPU emulation of tests/emu.
template <int NL, int BD, class C = SwimmerOneLane>
MZS_HD int swimmer_maze_step(const SwimmerDev& P, float* qf, float* vf, const float* action, int t_in, float* o, double* inner_reward,
                             double* info4, int* t_out, const C& cx = C()) {
  constexpr int NR = NL + 2, NH = NL - 1;
  double q[NR], v[NR], a[NH];
  for (int k = 0; k < NR; k++) { q[k] = (double)qf[k]; v[k] = (double)vf[k]; }
  for (int k = 0; k < NH; k++) a[k] = (double)action[k];
  int st = swimmer_env_step<NL>(cx, P, q, v, a, t_in, inner_reward, info4, t_out);
  bool badv = false;
  for (int k = 0; k < NR; k++) { badv = badv || !(fabs(q[k]) < 1e10) || !(fabs(v[k]) < 1e10); qf[k] = (float)q[k]; vf[k] = (float)v[k]; }
  if constexpr (BD > 0) {  // no contacts (swimmer.xml:3 collision="predefined"): drag, gravity on a z slide, joint limits
    double qb[BD], vb[BD];
    bool moving = false;
    for (int c = 0; c < BD; c++) { qb[c] = (double)qf[NR + c]; vb[c] = (double)vf[NR + c]; moving = moving || vb[c] != 0.0 || P.bd_axis[c] == 2; }
    if (moving) swimmer_block_step(P, qb, vb);
    for (int c = 0; c < BD; c++) { badv = badv || !(fabs(qb[c]) < 1e10) || !(fabs(vb[c]) < 1e10); qf[NR + c] = (float)qb[c]; vf[NR + c] = (float)vb[c]; }
  }
  if (badv) st |= MZ_STATUS_BAD_STATE;
  swimmer_obs_row<NL, BD>(P, qf, vf, *t_out, o);
  return st;
}
