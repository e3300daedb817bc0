/* mzo_env.c — CPU ORACLE (test infrastructure, not the product).
 *
 * Restates, in plain C / float64, the maze-level part of the reference hot
 * path; each function cites the reference lines it follows.  Pinned against the
 * golden vectors in tests/golden/ that were produced by running the reference's
 * own pure-Python modules (tests/golden/make_golden.py).
 *
 *   mzo_detect     <- CollisionDetector.detect + Line.*   maze_env_utils.py:96-123,186-206
 *   bounce logic   <- MazeEnv.step                         maze_env.py:451-464
 *   point pre-step <- PointEnv.step                        point.py:44-61
 *   ant reward     <- AntEnv.step / _forward_reward        ant.py:56-73
 *   mzo_env_obs    <- MazeEnv._get_obs + robot _get_obs    maze_env.py:351-369, ant.py:75-82, point.py:63-69
 *   mzo_task_eval  <- MazeGoal.neighbor, MazeTask.termination, reward variants
 *                                                          maze_task.py:43-47,77-81,110-111,403-407,592-604,619-621
 *   mzo_env_reset  <- reset_model                          ant.py:84-96, point.py:71-81
 */
#include <math.h>
#include <string.h>

#include "mzo.h"

uint64_t mzo_model_sizeof(void) { return sizeof(mz_model); }
uint64_t mzo_data_sizeof(void) { return sizeof(mzo_data); }

/* ---------------------------------------------------------------- RNG (shared definition with the kernels) */
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
uint32_t mzo_rng_u32(uint64_t seed, uint64_t env, uint32_t counter) {
  uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (env + 1));
  z = mix64(z + 0x9E3779B97F4A7C15ULL * ((uint64_t)counter + 1));
  return (uint32_t)(z >> 32);
}
static inline double u01(uint64_t seed, uint64_t env, uint32_t c) { /* [0,1) on a 2^-24 lattice */
  return (double)(mzo_rng_u32(seed, env, c) >> 8) * (1.0 / 16777216.0);
}
static inline double normal01(uint64_t seed, uint64_t env, uint32_t c) {
  double u1 = ((double)(mzo_rng_u32(seed, env, c) >> 8) + 1.0) * (1.0 / 16777216.0);
  double u2 = u01(seed, env, c + 1);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

/* ---------------------------------------------------------------- wall segments */
static inline double cross2(double ax, double ay, double bx, double by) { return ax * by + (-ay) * bx; }

int mzo_detect(const mz_model* m, const double* o, const double* n, double* point, double* reflection) {
  double mvx = n[0] - o[0], mvy = n[1] - o[1];
  if (hypot(mvx, mvy) <= 1e-8) return 0; /* maze_env_utils.py:189 */
  int found = 0, degenerate = 0;
  double best = 0;
  for (int k = 0; k < m->nseg; k++) {
    const double* s = m->seg[k];
    double wx = s[2] - s[0], wy = s[3] - s[1];
    /* wall._intersect(move): move's end points on both sides of the wall line (<= 0: touching counts) */
    double c1 = cross2(wx, wy, o[0] - s[0], o[1] - s[1]), c2 = cross2(wx, wy, n[0] - s[0], n[1] - s[1]);
    if (!(c1 * c2 <= 0.0)) continue;
    /* move._intersect(wall) */
    double c3 = cross2(mvx, mvy, s[0] - o[0], s[1] - o[1]), c4 = cross2(mvx, mvy, s[2] - o[0], s[3] - o[1]);
    if (!(c3 * c4 <= 0.0)) continue;
    /* wall._cross_point(move): other = move -> other.p1 + b / a * (other.p2 - other.p1) */
    double a = cross2(wx, wy, mvx, mvy), b = cross2(wx, wy, s[2] - o[0], s[3] - o[1]);
    if (a == 0.0) { degenerate = 1; continue; } /* the reference raises ZeroDivisionError here */
    double r = b / a, px = o[0] + r * mvx, py = o[1] + r * mvy;
    double dist = hypot(px - o[0], py - o[1]);
    if (!found || dist < best) {
      found = 1; best = dist;
      point[0] = px; point[1] = py;
      /* wall.reflection(new): p + 2 * (projection(p) - p), projection along -v1 (maze_env_utils.py:101-108) */
      double bx = -wx, by = -wy, n2 = hypot(bx, by);
      n2 = n2 * n2;
      double dx = n[0] - s[0], dy = n[1] - s[1];
      double sc = (dx * bx - (-dy) * by) / n2;
      double qx = s[0] + bx * sc, qy = s[1] + by * sc;
      reflection[0] = n[0] + 2.0 * (qx - n[0]);
      reflection[1] = n[1] + 2.0 * (qy - n[1]);
    }
  }
  if (degenerate && !found) return -1;
  return found;
}

/* Wall bounce of MazeEnv.step (maze_env.py:457-464): returns 0 no hit, 1 bounced, 2 gave up
 * (position restored to old), -1 when the reference would have raised (collinear). */
int mzo_bounce(const mz_model* m, const double* old_xy, const double* new_xy, double* final_xy) {
  double pt[2], rf[2];
  final_xy[0] = new_xy[0]; final_xy[1] = new_xy[1];
  int hit = mzo_detect(m, old_xy, new_xy, pt, rf);
  if (hit <= 0) return hit;
  double pos[2] = {pt[0] + m->restitution * (rf[0] - pt[0]), pt[1] + m->restitution * (rf[1] - pt[1])}, p2[2], r2[2];
  int again = mzo_detect(m, old_xy, pos, p2, r2);
  if (again < 0) return -1;
  if (again > 0) { final_xy[0] = old_xy[0]; final_xy[1] = old_xy[1]; return 2; }
  final_xy[0] = pos[0]; final_xy[1] = pos[1];
  return 1;
}

/* ---------------------------------------------------------------- task */
static int goal_neighbor(const mz_model* m, int g, const double* slot) {
  double s = 0;
  for (int k = 0; k < m->goal_dim[g]; k++) { double e = slot[k] - m->goal_pos[g][k]; s += e * e; }
  return sqrt(s) <= m->goal_threshold[g];
}
void mzo_task_eval(const mz_model* m, const double* obs, double* reward, int* done, int* goal_idx) {
  const double* rslot = m->reward_slot == MZ_SLOT_OBJECT ? obs + 3 : obs;
  const double* tslot = m->term_slot == MZ_SLOT_OBJECT ? obs + 3 : obs;
  int term = 0, first = -1, first_t = -1;
  for (int g = 0; g < m->ngoal; g++)
    if (goal_neighbor(m, g, tslot)) { term = 1; first_t = g; break; }
  for (int g = 0; g < m->ngoal; g++)
    if (goal_neighbor(m, g, rslot)) { first = g; break; }
  double r = 0.0;
  if (m->reward_kind == MZ_REWARD_FIRST_MATCH) {
    if (m->reward_binary) r = term ? 1.0 : m->penalty; /* maze_task.py:110-111 */
    else r = first >= 0 ? m->goal_reward_scale[first] : m->penalty; /* :403-407 */
  } else if (m->reward_kind == MZ_REWARD_NEG_DIST) {
    double s = 0;
    for (int k = 0; k < m->goal_dim[0]; k++) { double e = rslot[k] - m->goal_pos[0][k]; s += e * e; }
    r = -sqrt(s) / m->task_scale; /* :98-99, :619-621 */
  }
  *reward = r;
  *done = term;
  /* the goal that set the reward where the reward is a goal's; otherwise (zero / distance rewards) the first goal that
     terminates, on the slot termination() looks at */
  *goal_idx = m->reward_kind == MZ_REWARD_FIRST_MATCH ? first : first_t;
}

/* ---------------------------------------------------------------- top-down view (maze_env.py:262-349) */
/* Python's `x % 1` for floats (CPython float_rem): fmod, moved into [0, 1) */
static double py_mod1(double x) {
  double r = fmod(x, 1.0);
  if (r != 0.0) { if (r < 0.0) r += 1.0; }
  else r = 0.0;
  return r;
}

/* update_view (maze_env.py:268-320): a source at fractional grid coordinates (row, col) spreads a unit square over the
 * cell int(row), int(col) — Python's int() truncates towards zero while `% 1` floors, which is what the reference does
 * for negative coordinates too — and its eight neighbours */
static void view_splat(double* view, double row, double col, int d) {
  if (!(fabs(row) < 1e9) || !(fabs(col) < 1e9)) return;
  int r0 = (int)row, c0 = (int)col;
  double rf = py_mod1(row), cf = py_mod1(col);
  double wr[3] = {fmax(0.0, 0.5 - rf), fmin(1.0, rf + 0.5) - fmax(0.0, rf - 0.5), fmax(0.0, rf - 0.5)};
  double wc[3] = {fmax(0.0, 0.5 - cf), fmin(1.0, cf + 0.5) - fmax(0.0, cf - 0.5), fmax(0.0, cf - 0.5)};
  for (int a = -1; a <= 1; a++)
    for (int b = -1; b <= 1; b++) {
      int r = r0 + a, c = c0 + b;
      if (r >= 0 && r < 5 && c >= 0 && c < 5) view[(r * 5 + c) * 3 + d] += wr[a + 1] * wc[b + 1];
    }
}

/* get_top_down_view: walls -> channel 0, chasms -> channel 1, movable blocks (body positions, creation order) -> channel 2,
 * all relative to the torso position (robot_x, robot_y); `_xy_to_rowcol` (maze_env.py:90-93) puts the robot at (2.5, 2.5) */
void mzo_top_down_view(const mz_model* m, double robot_x, double robot_y, int nblock, const double* block_xy, double* view) {
  const double sc = m->maze_scale;
  for (int k = 0; k < MZ_VIEW_DIM; k++) view[k] = 0.0;
  for (int i = 0; i < m->grid_rows; i++)
    for (int j = 0; j < m->grid_cols; j++) {
      int cell = m->grid[i][j];
      if (cell != MZ_CELL_BLOCK && cell != MZ_CELL_CHASM) continue;
      double x = (double)j * sc - m->torso_x, y = (double)i * sc - m->torso_y;
      x = x - robot_x;
      y = y - robot_y;
      view_splat(view, 2.0 + (y + sc / 2.0) / sc, 2.0 + (x + sc / 2.0) / sc, cell == MZ_CELL_BLOCK ? 0 : 1);
    }
  for (int b = 0; b < nblock; b++) {
    double x = block_xy[2 * b] - robot_x, y = block_xy[2 * b + 1] - robot_y;
    view_splat(view, 2.0 + (y + sc / 2.0) / sc, 2.0 + (x + sc / 2.0) / sc, 2);
  }
}

/* ---------------------------------------------------------------- obs */
static void body_origin(const mz_model* m, const mzo_env_state* s, int body, double* p) {
  if (m->body_jntnum[body] == 1 && m->jnt_type[m->body_jntadr[body]] == MZ_JNT_FREE) { /* free-joint object ball: qpos holds the pose */
    for (int c = 0; c < 3; c++) p[c] = s->qpos[m->jnt_qposadr[m->body_jntadr[body]] + c];
    return;
  }
  for (int c = 0; c < 3; c++) p[c] = m->body_pos[body][c];
  for (int j = m->body_jntadr[body]; j < m->body_jntadr[body] + m->body_jntnum[body]; j++)
    if (m->jnt_type[j] == MZ_JNT_SLIDE) {
      double q = s->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
      for (int c = 0; c < 3; c++) p[c] += m->jnt_axis[j][c] * q;
    }
}

void mzo_env_obs(const mz_model* m, const mzo_env_state* s, double* obs) {
  int k = 0;
  for (int i = 0; i < 3; i++) obs[k++] = s->qpos[i];
  /* get_body_com = the body's frame origin: spawn position + its slide coordinates along their axes (maze_env.py:360-368);
   * balls come before blocks; a ball's z stays 0 (hinge flavour: x / y slides and a z hinge) */
  for (int pass = 0; pass < 2; pass++) {
    int n = pass == 0 ? (m->observe_balls ? m->nball : 0) : (m->observe_blocks ? m->nblock : 0);
    for (int b = 0; b < n; b++) {
      double p[3];
      body_origin(m, s, pass == 0 ? m->ball_bodyid[b] : m->block_bodyid[b], p);
      for (int c = 0; c < 3; c++) obs[k++] = p[c];
    }
  }
  for (int i = 3; i < m->nq_robot; i++) obs[k++] = s->qpos[i];
  for (int i = 0; i < m->nv_robot; i++) obs[k++] = s->qvel[i];
  if (m->top_down_view) { /* maze_env.py:353-354,369: the view sits before the time entry; torso position = qpos[:2] of every robot */
    double bxy[8], p[3];
    for (int b = 0; b < m->nblock; b++) { body_origin(m, s, m->block_bodyid[b], p); bxy[2 * b] = p[0]; bxy[2 * b + 1] = p[1]; }
    mzo_top_down_view(m, s->qpos[0], s->qpos[1], m->nblock, bxy, obs + k);
    k += MZ_VIEW_DIM;
  }
  obs[k++] = s->t * 0.001; /* maze_env.py:369 */
}

/* ---------------------------------------------------------------- reset */
void mzo_env_reset(const mz_model* m, mzo_env_state* s, uint64_t seed, uint64_t env) {
  memset(s, 0, sizeof(*s));
  for (int i = 0; i < m->nq; i++) {
    s->qpos[i] = m->qpos0[i];
    if (i < m->nq_robot) s->qpos[i] += -0.1 + 0.2 * u01(seed, env, (uint32_t)i);
  }
  if (m->njnt > 0 && m->jnt_type[0] == MZ_JNT_FREE) { /* set_state -> mj_forward: mj_kinematics normalises the quaternion in qpos [ASSUME-8] */
    double* q = s->qpos + m->jnt_qposadr[0] + 3;
    double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; k++) q[k] /= nn;
  }
  for (int i = 0; i < m->nv_robot; i++) {
    uint32_t c = (uint32_t)(m->nq + 2 * i);
    if (m->reset_qvel_kind == 0) s->qvel[i] = 0.1 * normal01(seed, env, c);
    else if (m->reset_qvel_kind == 1) s->qvel[i] = 0.1 * u01(seed, env, c);
    else s->qvel[i] = -0.1 + 0.2 * u01(seed, env, c);
  }
}

/* ---------------------------------------------------------------- MazeEnv.step */
void mzo_env_step(const mz_model* m, mzo_env_state* s, const double* action, double* obs, double* reward, uint8_t* done,
                  int32_t* goal_idx, double* info4, double solver_tol) {
  mzo_data d;
  memset(&d, 0, sizeof(d));
  d.solver_tol = solver_tol;
  memcpy(d.qpos, s->qpos, sizeof(double) * m->nq);
  memcpy(d.qvel, s->qvel, sizeof(double) * m->nv);
  memcpy(d.qacc_warmstart, s->warmstart, sizeof(double) * m->nv);
  s->t += 1; /* maze_env.py:449 */
  double old_xy[2] = {d.qpos[0], d.qpos[1]};
  double inner = 0.0, fwd = 0.0, ctrl_cost = 0.0;
  if (m->step_kind ? m->step_kind == 2 : m->robot == MZ_ROBOT_POINT) { /* mz_model.step_kind: 0 = the robot family's, 1 motors, 2 the Point's */
    /* point.py:45-59 */
    double th = d.qpos[2] + action[1];
    if (th < -M_PI) th += M_PI * 2;
    else if (M_PI < th) th -= M_PI * 2;
    d.qpos[2] = th;
    d.qpos[0] += cos(th) * action[0];
    d.qpos[1] += sin(th) * action[0];
    for (int i = 0; i < m->nv; i++) {
      if (d.qvel[i] < -m->velocity_limit) d.qvel[i] = -m->velocity_limit;
      if (d.qvel[i] > m->velocity_limit) d.qvel[i] = m->velocity_limit;
    }
    for (int k = 0; k < m->frame_skip; k++) mzo_mj_step(m, &d, NULL); /* ctrl is never written */
  } else {
    /* ant.py:61-73 / swimmer.py:37-48 */
    for (int k = 0; k < m->frame_skip; k++) mzo_mj_step(m, &d, action);
    double dt = m->timestep * m->frame_skip;
    double vx = (d.qpos[0] - old_xy[0]) / dt, vy = (d.qpos[1] - old_xy[1]) / dt;
    fwd = sqrt(vx * vx + vy * vy);
    for (int a = 0; a < m->nu; a++) ctrl_cost += action[a] * action[a];
    ctrl_cost *= m->ctrl_cost_weight;
    inner = m->forward_reward_weight * fwd - ctrl_cost;
  }
  if (m->manual_collision) { /* maze_env.py:451-464 */
    double new_xy[2] = {d.qpos[0], d.qpos[1]}, fin[2];
    int r = mzo_bounce(m, old_xy, new_xy, fin);
    if (r < 0) s->status |= 8;
    d.qpos[0] = fin[0]; d.qpos[1] = fin[1];
  }
  memcpy(s->qpos, d.qpos, sizeof(double) * m->nq);
  memcpy(s->qvel, d.qvel, sizeof(double) * m->nv);
  memcpy(s->warmstart, d.qacc_warmstart, sizeof(double) * m->nv);
  s->status |= d.status;
  mzo_env_obs(m, s, obs);
  double outer; int term, gi;
  mzo_task_eval(m, obs, &outer, &term, &gi);
  *reward = m->inner_reward_scaling * inner + outer; /* maze_env.py:477-481 */
  *done = (uint8_t)((term ? 1 : 0) | (s->t >= m->max_episode_steps ? 2 : 0));
  if (goal_idx) *goal_idx = gi;
  if (info4) { info4[0] = d.qpos[0]; info4[1] = d.qpos[1]; info4[2] = fwd; info4[3] = -ctrl_cost; }
}

/* ---------------------------------------------------------------- batch drivers */
void mzo_batch_step(const mz_model* m, int n, double* qpos, double* qvel, double* warm, int32_t* t, const double* actions,
                    double* obs, double* reward, uint8_t* done, int32_t* goal_idx, double* info, int32_t* status,
                    int nthreads, double solver_tol) {
  (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
  for (int e = 0; e < n; e++) {
    mzo_env_state s;
    memset(&s, 0, sizeof(s));
    memcpy(s.qpos, qpos + (size_t)e * m->nq, sizeof(double) * m->nq);
    memcpy(s.qvel, qvel + (size_t)e * m->nv, sizeof(double) * m->nv);
    memcpy(s.warmstart, warm + (size_t)e * m->nv, sizeof(double) * m->nv);
    s.t = t[e];
    double inf[4];
    int32_t gi;
    mzo_env_step(m, &s, actions + (size_t)e * m->nu, obs + (size_t)e * m->obs_dim, reward + e, done + e, &gi, inf, solver_tol);
    memcpy(qpos + (size_t)e * m->nq, s.qpos, sizeof(double) * m->nq);
    memcpy(qvel + (size_t)e * m->nv, s.qvel, sizeof(double) * m->nv);
    memcpy(warm + (size_t)e * m->nv, s.warmstart, sizeof(double) * m->nv);
    t[e] = s.t;
    if (goal_idx) goal_idx[e] = gi;
    if (info) memcpy(info + (size_t)e * 4, inf, sizeof(inf));
    if (status) status[e] = s.status;
  }
}

void mzo_batch_reset(const mz_model* m, int n, const uint8_t* mask, uint64_t seed, uint64_t env0, double* qpos, double* qvel,
                     double* warm, int32_t* t, double* obs) {
  for (int e = 0; e < n; e++) {
    if (mask && !mask[e]) continue;
    mzo_env_state s;
    mzo_env_reset(m, &s, seed, env0 + (uint64_t)e);
    memcpy(qpos + (size_t)e * m->nq, s.qpos, sizeof(double) * m->nq);
    memcpy(qvel + (size_t)e * m->nv, s.qvel, sizeof(double) * m->nv);
    memset(warm + (size_t)e * m->nv, 0, sizeof(double) * m->nv);
    t[e] = 0;
    if (obs) mzo_env_obs(m, &s, obs + (size_t)e * m->obs_dim);
  }
}

void mzo_batch_forward(const mz_model* m, int n, const double* qpos, const double* qvel, const double* warm,
                       const double* actions, double* qacc, int32_t* counts, double* Mout, double* bias) {
  for (int e = 0; e < n; e++) {
    mzo_data d;
    memset(&d, 0, sizeof(d));
    memcpy(d.qpos, qpos + (size_t)e * m->nq, sizeof(double) * m->nq);
    memcpy(d.qvel, qvel + (size_t)e * m->nv, sizeof(double) * m->nv);
    if (warm) memcpy(d.qacc_warmstart, warm + (size_t)e * m->nv, sizeof(double) * m->nv);
    mzo_forward(m, &d, actions ? actions + (size_t)e * m->nu : NULL);
    memcpy(qacc + (size_t)e * m->nv, d.qacc, sizeof(double) * m->nv);
    if (counts) { counts[2 * e] = d.ncon; counts[2 * e + 1] = d.nefc; }
    if (Mout)
      for (int i = 0; i < m->nv; i++)
        for (int j = 0; j < m->nv; j++) Mout[((size_t)e * m->nv + i) * m->nv + j] = d.M[i][j];
    if (bias) memcpy(bias + (size_t)e * m->nv, d.qfrc_bias, sizeof(double) * m->nv);
  }
}

/* diagnostics for the invariants tests: report = [ncon, nefc, solver_iter, kkt residual (inf-norm of
 * M(qacc - qacc_smooth) - J^T f), sum of row forces, min row force, max |f * max(jar,0)| complementarity, status] */
void mzo_forward_report(const mz_model* m, const double* qpos, const double* qvel, const double* warm, const double* ctrl,
                        double* report, double* qacc_out) {
  mzo_data d;
  memset(&d, 0, sizeof(d));
  memcpy(d.qpos, qpos, sizeof(double) * m->nq);
  memcpy(d.qvel, qvel, sizeof(double) * m->nv);
  if (warm) memcpy(d.qacc_warmstart, warm, sizeof(double) * m->nv);
  mzo_forward(m, &d, ctrl);
  double res = 0, fsum = 0, fmin = 0, comp = 0;
  for (int i = 0; i < m->nv; i++) {
    double s = 0;
    for (int j = 0; j < m->nv; j++) s += d.M[i][j] * (d.qacc[j] - d.qacc_smooth[j]);
    for (int r = 0; r < d.nefc; r++) s -= d.efc_J[r][i] * d.efc_force[r];
    if (fabs(s) > res) res = fabs(s);
  }
  for (int r = 0; r < d.nefc; r++) {
    double jar = -d.efc_aref[r];
    for (int i = 0; i < m->nv; i++) jar += d.efc_J[r][i] * d.qacc[i];
    fsum += d.efc_force[r];
    if (r == 0 || d.efc_force[r] < fmin) fmin = d.efc_force[r];
    double c = fabs(d.efc_force[r] * (jar > 0 ? jar : 0));
    if (c > comp) comp = c;
    /* soft-constraint law: f = -D * jar on active rows */
    if (jar < 0) { double e = fabs(d.efc_force[r] + d.efc_D[r] * jar); if (e > comp) comp = e; }
  }
  report[0] = d.ncon; report[1] = d.nefc; report[2] = d.solver_iter; report[3] = res; report[4] = fsum; report[5] = fmin;
  report[6] = comp; report[7] = d.status;
  if (qacc_out) memcpy(qacc_out, d.qacc, sizeof(double) * m->nv);
}

/* contact set of one configuration (parity tests, tests/test_mujoco_crosscheck.py): per contact 10 doubles
 * dist | pos[3] | normal[3] (geom1 -> geom2) | geom1 | geom2 (-1: an implicit maze box) | active (dist < margin - gap);
 * returns the number of contacts (at most max_con rows are written) */
int mzo_contacts(const mz_model* m, const double* qpos, int max_con, double* out) {
  mzo_data d;
  memset(&d, 0, sizeof(d));
  memcpy(d.qpos, qpos, sizeof(double) * m->nq);
  mzo_forward(m, &d, NULL);
  for (int c = 0; c < d.ncon && c < max_con; c++) {
    double* o = out + 10 * c;
    o[0] = d.con[c].dist;
    for (int k = 0; k < 3; k++) { o[1 + k] = d.con[c].pos[k]; o[4 + k] = d.con[c].frame[k]; }
    o[7] = d.con[c].geom1; o[8] = d.con[c].geom2; o[9] = d.con[c].dist < d.con[c].includemargin;
  }
  return d.ncon;
}

/* raw mj_step x n on one state (no maze logic) + energy, for invariants tests */
void mzo_raw_steps(const mz_model* m, double* qpos, double* qvel, const double* ctrl, int nsteps, double* energy_out) {
  mzo_data d;
  memset(&d, 0, sizeof(d));
  memcpy(d.qpos, qpos, sizeof(double) * m->nq);
  memcpy(d.qvel, qvel, sizeof(double) * m->nv);
  for (int k = 0; k < nsteps; k++) {
    if (energy_out) energy_out[k] = mzo_energy(m, &d);
    mzo_mj_step(m, &d, ctrl);
  }
  memcpy(qpos, d.qpos, sizeof(double) * m->nq);
  memcpy(qvel, d.qvel, sizeof(double) * m->nv);
}

/* diagnostics: list the contacts of one state: out[c] = (geom1, geom2, dist, pos[3], normal[3]) */
int mzo_list_contacts(const mz_model* m, const double* qpos, const double* qvel, double* out, int maxc) {
  mzo_data d;
  memset(&d, 0, sizeof(d));
  memcpy(d.qpos, qpos, sizeof(double) * m->nq);
  memcpy(d.qvel, qvel, sizeof(double) * m->nv);
  mzo_forward(m, &d, NULL);
  for (int c = 0; c < d.ncon && c < maxc; c++) {
    double* o = out + 9 * c;
    o[0] = d.con[c].geom1; o[1] = d.con[c].geom2; o[2] = d.con[c].dist;
    for (int k = 0; k < 3; k++) { o[3 + k] = d.con[c].pos[k]; o[6 + k] = d.con[c].frame[k]; }
  }
  return d.ncon;
}
