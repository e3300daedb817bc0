// ant_emu.cpp — CPU emulation of the Ant kernel's lane-group code (TEST INFRASTRUCTURE).
//
// Compiles mujoco_maze_amd/csrc/ant_dyn.h — the exact source the HIP kernel
// instantiates — with the one-lane HostCtx, so that the kernel's *logic and fp32
// numerics* can be compared with the float64 oracle on a machine without a GPU.
// It is not a fallback: nothing under mujoco_maze_amd/ loads this library and the
// product path raises if the HIP library / a GPU is missing.
#include <stdlib.h>

#include "../../mujoco_maze_amd/csrc/ant_dyn.h"

template <int NB>
static int env_step_t(const AntDev& K, int n, float* qpos, float* qvel, float* warm, int32_t* t, const float* actions, float* obs,
                      float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status, int32_t* iters) {
  using D = AntDims<NB>;
  HostCtx cx;
  AntScratchT<NB>* s = (AntScratchT<NB>*)calloc(1, sizeof(AntScratchT<NB>));
  const int obs_dim = ANT_OBS + ant_obs_extra<NB>(K);
  for (int e = 0; e < n; e++) {
    for (int k = 0; k < D::NQ; k++) s->qpos[k] = qpos[e * D::NQ + k];
    for (int k = 0; k < D::NV; k++) { s->qvel[k] = qvel[e * D::NV + k]; s->warm[k] = warm[e * D::NV + k]; }
    int gi = -1, tout = t[e];
    s->prof[15] = 0;
    ant_env_step<NB>(cx, K, *s, actions + e * ANT_NU, obs + e * obs_dim, reward + e, done + e, &gi, info ? info + 4 * e : nullptr, &tout);
    for (int k = 0; k < D::NQ; k++) qpos[e * D::NQ + k] = s->qpos[k];
    for (int k = 0; k < D::NV; k++) { qvel[e * D::NV + k] = s->qvel[k]; warm[e * D::NV + k] = s->warm[k]; }
    t[e] = tout;
    if (goal_idx) goal_idx[e] = gi;
    if (status) status[e] = s->status;
    if (iters) iters[e] = (int32_t)s->prof[15];  // Newton iterations summed over the step's forward evaluations
  }
  free(s);
  return MZ_OK;
}

template <int NB>
static int forward_t(const AntDev& K, int n, const float* qpos, const float* qvel, const float* warm, const float* actions,
                     float* qacc, int32_t* counts, float* Mout, float* bias, float* qas) {
  using D = AntDims<NB>;
  HostCtx cx;
  AntScratchT<NB>* s = (AntScratchT<NB>*)calloc(1, sizeof(AntScratchT<NB>));
  for (int e = 0; e < n; e++) {
    for (int k = 0; k < D::NQ; k++) s->qpos[k] = qpos[e * D::NQ + k];
    for (int k = 0; k < D::NV; k++) { s->qvel[k] = qvel[e * D::NV + k]; s->warm[k] = warm ? warm[e * D::NV + k] : 0.f; s->fact[k] = 0.f; }
    if (actions)
      for (int u = 0; u < ANT_NU; u++) s->fact[K.act_dof[u]] = K.gear * fminf(fmaxf(actions[e * ANT_NU + u], K.ctrl_lo), K.ctrl_hi);
    s->status = 0;
    ant_fill_tables<NB>(cx, K, *s);
    ant_forward<NB>(cx, K, *s, true);
    for (int k = 0; k < D::NV; k++) qacc[e * D::NV + k] = s->qacc[k];
    if (counts) { counts[2 * e] = s->ncon; counts[2 * e + 1] = s->iters; }
    if (bias) for (int k = 0; k < D::NV; k++) bias[e * D::NV + k] = s->bias[k];
    if (qas) for (int k = 0; k < D::NV; k++) qas[e * D::NV + k] = s->qas[k];
    if (Mout) {
      float* M = Mout + (size_t)e * D::NV * D::NV;
      float x[D::NV];
      for (int j = 0; j < D::NV; j++) {
        for (int k = 0; k < D::NV; k++) x[k] = k == j ? 1.f : 0.f;
        for (int i = 0; i < D::NV; i++) M[i * D::NV + j] = arrow_row_mul<D::NH>(s->M, x, i);
      }
    }
  }
  free(s);
  return MZ_OK;
}

static int make_dev(AntDev* K, const mz_model* m, int max_iter, float tol, float rtol) {
  char err[128];
  int rc = ant_dev_from_model(K, m, err, sizeof(err));
  if (rc != MZ_OK) return rc;
  if (max_iter > 0) K->max_iter = max_iter;
  if (tol > 0) K->tol = tol;
  if (rtol >= 0) K->rtol = rtol;
  return MZ_OK;
}

#include "../../mujoco_maze_amd/csrc/mz_view.h"

extern "C" {


// One MazeEnv.step for n envs (row-major arrays as in the C-ABI's get/set_state).
int emu_ant_env_step(const mz_model* m, int n, float* qpos, float* qvel, float* warm, int32_t* t, const float* actions,
                     float* obs, float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status,
                     int32_t* iters, int max_iter, float tol, float rtol) {
  AntDev K;
  int rc = make_dev(&K, m, max_iter, tol, rtol);
  if (rc != MZ_OK) return rc;
  switch (K.nball ? 5 : (K.nblock == 1 && K.block_nax == 3 ? 4 : K.nblock)) {
    case 5: return env_step_t<5>(K, n, qpos, qvel, warm, t, actions, obs, reward, done, goal_idx, info, status, iters);
    case 4: return env_step_t<4>(K, n, qpos, qvel, warm, t, actions, obs, reward, done, goal_idx, info, status, iters);
    case 0: return env_step_t<0>(K, n, qpos, qvel, warm, t, actions, obs, reward, done, goal_idx, info, status, iters);
    case 1: return env_step_t<1>(K, n, qpos, qvel, warm, t, actions, obs, reward, done, goal_idx, info, status, iters);
    case 2: return env_step_t<2>(K, n, qpos, qvel, warm, t, actions, obs, reward, done, goal_idx, info, status, iters);
    case 3: return env_step_t<3>(K, n, qpos, qvel, warm, t, actions, obs, reward, done, goal_idx, info, status, iters);
    default: return MZ_ERR_UNSUPPORTED;
  }
}

// One forward-dynamics evaluation per env: qacc [n,nv], counts [n,2] = (ncon, newton iterations),
// optional dense mass matrix [n,nv,nv], bias [n,nv], qacc_smooth [n,nv].
int emu_ant_forward(const mz_model* m, int n, const float* qpos, const float* qvel, const float* warm, const float* actions,
                    float* qacc, int32_t* counts, float* Mout, float* bias, float* qas, int max_iter, float tol, float rtol) {
  AntDev K;
  int rc = make_dev(&K, m, max_iter, tol, rtol);
  if (rc != MZ_OK) return rc;
  switch (K.nball ? 5 : (K.nblock == 1 && K.block_nax == 3 ? 4 : K.nblock)) {
    case 5: return forward_t<5>(K, n, qpos, qvel, warm, actions, qacc, counts, Mout, bias, qas);
    case 4: return forward_t<4>(K, n, qpos, qvel, warm, actions, qacc, counts, Mout, bias, qas);
    case 0: return forward_t<0>(K, n, qpos, qvel, warm, actions, qacc, counts, Mout, bias, qas);
    case 1: return forward_t<1>(K, n, qpos, qvel, warm, actions, qacc, counts, Mout, bias, qas);
    case 2: return forward_t<2>(K, n, qpos, qvel, warm, actions, qacc, counts, Mout, bias, qas);
    case 3: return forward_t<3>(K, n, qpos, qvel, warm, actions, qacc, counts, Mout, bias, qas);
    default: return MZ_ERR_UNSUPPORTED;
  }
}
}

// ---------------------------------------------------------------- Point (+ movable blocks): the lane-group step of planar_step_kernel
#include "../../mujoco_maze_amd/csrc/planar_dyn.h"

template <int NB, int NS>
static int planar_env_step_t(const PointDev& P, int n, float* qpos, float* qvel, int32_t* t, const float* actions, float* obs,
                             float* reward, uint8_t* done, int32_t* goal_idx, int32_t* status) {
  using D = PlanarDims<NB, NS>;
  constexpr int NV = D::NV, NOBS = D::NOBS;
  HostCtx cx;
  PlanarScratch<NB, NS>* s = (PlanarScratch<NB, NS>*)calloc(1, sizeof(PlanarScratch<NB, NS>));
  for (int e = 0; e < n; e++) {
    double a[2] = {(double)actions[2 * e], (double)actions[2 * e + 1]};
    for (int k = 0; k < NV; k++) { s->q[k] = (double)qpos[NV * e + k]; s->v[k] = (double)qvel[NV * e + k]; }
    if constexpr (NB == 0 && NS == 0) { s->wds[0] = s->wds[1] = s->wds[2] = 0.0; }  // (a state handed in carries no warm start: as after mz_set_state)
    const int t_new = t[e] + 1;
    planar_env_step<NB, NS>(cx, P, *s, a);
    float o[NOBS];
    for (int i = 0; i < NOBS; i++) o[i] = planar_obs_elem<NB, NS>(P, *s, i, t_new);
    float outer; int tm, gi;
    task_eval_dev(P.task, o, &outer, &tm, &gi);
    for (int k = 0; k < NOBS; k++) obs[NOBS * e + k] = o[k];
    for (int k = 0; k < NV; k++) { qpos[NV * e + k] = (float)s->q[k]; qvel[NV * e + k] = (float)s->v[k]; }
    reward[e] = outer;
    done[e] = (uint8_t)((tm ? 1 : 0) | (t_new >= P.task.max_steps ? 2 : 0));
    if (goal_idx) goal_idx[e] = gi;
    if (status) status[e] = s->status;
    t[e] = t_new;
  }
  free(s);
  return MZ_OK;
}

extern "C" int emu_point_env_step(const mz_model* m, int n, float* qpos, float* qvel, int32_t* t, const float* actions, float* obs,
                                  float* reward, uint8_t* done, int32_t* goal_idx, int32_t* status) {
  PointDev* P = (PointDev*)calloc(1, sizeof(PointDev));
  char err[128];
  int rc = point_dev_from_model(P, m, err, sizeof(err));
  if (rc != MZ_OK) { free(P); return rc; }
  if (P->nball) rc = planar_env_step_t<0, 1>(*P, n, qpos, qvel, t, actions, obs, reward, done, goal_idx, status);
  else switch (P->nblock) {
    case 0: rc = planar_env_step_t<0, 0>(*P, n, qpos, qvel, t, actions, obs, reward, done, goal_idx, status); break;
    case 1: rc = planar_env_step_t<1, 0>(*P, n, qpos, qvel, t, actions, obs, reward, done, goal_idx, status); break;
    case 2: rc = planar_env_step_t<2, 0>(*P, n, qpos, qvel, t, actions, obs, reward, done, goal_idx, status); break;
    default: rc = planar_env_step_t<3, 0>(*P, n, qpos, qvel, t, actions, obs, reward, done, goal_idx, status); break;
  }
  free(P);
  return rc;
}

extern "C" void emu_pt_sincos(int n, const double* x, double* s, double* c) {
  for (int i = 0; i < n; i++) pt_sincos(x[i], s + i, c + i);
}

// ---------------------------------------------------------------- Swimmer: the per-lane step function of swimmer_step_kernel
#include "../../mujoco_maze_amd/csrc/swimmer_dyn.h"

template <int NL, int NB>
static int swimmer_env_step_t(const SwimmerDev& P, int n, float* qpos, float* qvel, int32_t* t, const float* actions, float* obs,
                              float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status) {
  constexpr int NR = NL + 2, NV = NR + NB, NH = NL - 1;  // NB: slide dofs of the movable block (0, 2, 3)
  const int nb3 = (NB && P.observe_blocks) ? 3 : 0, NO = 2 * NV + 1 + nb3;
  for (int e = 0; e < n; e++) {
    float o[2 * NV + 4];
    double inner, inf4[4];
    int t_new;
    int st = swimmer_maze_step<NL, NB>(P, qpos + NV * e, qvel + NV * e, actions + NH * e, t[e], o, &inner, inf4, &t_new);
    float outer; int tm, gi;
    task_eval_dev(P.task, o, &outer, &tm, &gi);
    for (int k = 0; k < NO; k++) obs[NO * e + k] = o[k];
    reward[e] = (float)(P.task.inner_scale * inner) + outer;
    done[e] = (uint8_t)((tm ? 1 : 0) | (t_new >= P.task.max_steps ? 2 : 0));
    if (goal_idx) goal_idx[e] = gi;
    if (info) for (int k = 0; k < 4; k++) info[4 * e + k] = (float)inf4[k];
    if (status) status[e] = st;
    t[e] = t_new;
  }
  return MZ_OK;
}

extern "C" int emu_swimmer_env_step(const mz_model* m, int n, float* qpos, float* qvel, int32_t* t, const float* actions, float* obs,
                                    float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status) {
  SwimmerDev P;
  char err[128];
  int rc = swimmer_dev_from_model(&P, m, err, sizeof(err));
  if (rc != MZ_OK) return rc;
  const int bd = P.nblock ? P.nbdof : 0;
#define MZ_SW(NL, BD) swimmer_env_step_t<NL, BD>(P, n, qpos, qvel, t, actions, obs, reward, done, goal_idx, info, status)
  if (P.nlink == 3) return bd == 3 ? MZ_SW(3, 3) : (bd == 2 ? MZ_SW(3, 2) : MZ_SW(3, 0));
  if (P.nlink == 2) return bd == 3 ? MZ_SW(2, 3) : (bd == 2 ? MZ_SW(2, 2) : MZ_SW(2, 0));
  if (P.nlink == 4) return MZ_SW(4, 0);
  if (P.nlink == 5) return MZ_SW(5, 0);
  return MZ_SW(6, 0);
#undef MZ_SW
}

// ---------------------------------------------------------------- Swimmer, lane-group form: the code path swimmer_step_kernel runs
// The kernel advances one env on G adjacent lanes (lane b = link b; swimmer_dyn.h, `C::nlanes > 1` branches: per-lane link
// constants, group sums, the slides' constant corner of M, the structured factor of the longer chains) — a different branch of
// swimmer_forward than the one-lane form above.  Here the G lanes of a group are G host threads in lock step: a group sum /
// a read of another lane's value is a slot array between two barriers, added up in the order of the device's DPP butterfly
// (pairs, pairs of pairs, ...).  Slow, exact in structure; test infrastructure only.
#include <pthread.h>
template <int G>
struct LaneGroupShared { pthread_barrier_t bar; double slot[G]; };
template <int G>
struct SwimmerLanesEmu {
  static constexpr int nlanes = G;
  int l;
  LaneGroupShared<G>* sh;
  int lane0() const { return l; }
  double gsum(double x) const {
    sh->slot[l] = x;
    pthread_barrier_wait(&sh->bar);
    double t[G];
    for (int i = 0; i < G; i++) t[i] = sh->slot[i];
    for (int w = 1; w < G; w *= 2)
      for (int i = 0; i < G; i += 2 * w) t[i] = t[i] + t[i + w];
    pthread_barrier_wait(&sh->bar);
    return t[0];
  }
  double from_lane(double x, int k) const {
    sh->slot[l] = x;
    pthread_barrier_wait(&sh->bar);
    const double r = sh->slot[k];
    pthread_barrier_wait(&sh->bar);
    return r;
  }
  void stamp(int) const {}
};
template <int NL, int NB, int G>
struct SwimmerLaneJob {
  const SwimmerDev* P; LaneGroupShared<G>* sh; int lane;
  const float *q_in, *v_in, *act; int t_in;
  float q[NL + 2 + NB], v[NL + 2 + NB], o[2 * (NL + 2 + NB) + 4];
  double inner, inf4[4]; int t_new, st;
  static void* run(void* arg) {
    SwimmerLaneJob& j = *(SwimmerLaneJob*)arg;
    for (int k = 0; k < NL + 2 + NB; k++) { j.q[k] = j.q_in[k]; j.v[k] = j.v_in[k]; }
    SwimmerLanesEmu<G> cx{j.lane, j.sh};
    j.st = swimmer_maze_step<NL, NB, SwimmerLanesEmu<G>>(*j.P, j.q, j.v, j.act, j.t_in, j.o, &j.inner, j.inf4, &j.t_new, cx);
    return nullptr;
  }
};
template <int NL, int NB>
static int swimmer_env_step_lanes_t(const SwimmerDev& P, int n, float* qpos, float* qvel, int32_t* t, const float* actions, float* obs,
                                    float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status) {
  constexpr int G = NL <= 4 ? 4 : 8;  // as mzk_planar_step launches it
  constexpr int NR = NL + 2, NV = NR + NB, NH = NL - 1;
  const int nb3 = (NB && P.observe_blocks) ? 3 : 0, NO = 2 * NV + 1 + nb3;
  LaneGroupShared<G> sh;
  pthread_barrier_init(&sh.bar, nullptr, G);
  for (int e = 0; e < n; e++) {
    SwimmerLaneJob<NL, NB, G> job[G];
    pthread_t th[G];
    for (int l = 0; l < G; l++) {
      job[l].P = &P; job[l].sh = &sh; job[l].lane = l; job[l].q_in = qpos + NV * e; job[l].v_in = qvel + NV * e; job[l].act = actions + NH * e; job[l].t_in = t[e];
      pthread_create(&th[l], nullptr, SwimmerLaneJob<NL, NB, G>::run, &job[l]);
    }
    for (int l = 0; l < G; l++) pthread_join(th[l], nullptr);
    // every lane of the group ends with the same state (the kernel stores lane 0's): part of what this emulation checks
    for (int l = 1; l < G; l++)
      for (int k = 0; k < NV; k++) if (memcmp(&job[l].q[k], &job[0].q[k], 4) || memcmp(&job[l].v[k], &job[0].v[k], 4)) { pthread_barrier_destroy(&sh.bar); return MZ_ERR_ARG; }
    const SwimmerLaneJob<NL, NB, G>& j = job[0];
    for (int k = 0; k < NV; k++) { qpos[NV * e + k] = j.q[k]; qvel[NV * e + k] = j.v[k]; }
    float outer; int tm, gi;
    task_eval_dev(P.task, j.o, &outer, &tm, &gi);
    for (int k = 0; k < NO; k++) obs[NO * e + k] = j.o[k];
    reward[e] = (float)(P.task.inner_scale * j.inner) + outer;
    done[e] = (uint8_t)((tm ? 1 : 0) | (j.t_new >= P.task.max_steps ? 2 : 0));
    if (goal_idx) goal_idx[e] = gi;
    if (info) for (int k = 0; k < 4; k++) info[4 * e + k] = (float)j.inf4[k];
    if (status) status[e] = j.st;
    t[e] = j.t_new;
  }
  pthread_barrier_destroy(&sh.bar);
  return MZ_OK;
}
extern "C" int emu_swimmer_env_step_lanes(const mz_model* m, int n, float* qpos, float* qvel, int32_t* t, const float* actions, float* obs,
                                          float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status) {
  SwimmerDev P;
  char err[128];
  int rc = swimmer_dev_from_model(&P, m, err, sizeof(err));
  if (rc != MZ_OK) return rc;
  const int bd = P.nblock ? P.nbdof : 0;
#define MZ_SW(NL, BD) swimmer_env_step_lanes_t<NL, BD>(P, n, qpos, qvel, t, actions, obs, reward, done, goal_idx, info, status)
  if (P.nlink == 3) return bd == 3 ? MZ_SW(3, 3) : (bd == 2 ? MZ_SW(3, 2) : MZ_SW(3, 0));
  if (P.nlink == 2) return bd == 3 ? MZ_SW(2, 3) : (bd == 2 ? MZ_SW(2, 2) : MZ_SW(2, 0));
  if (P.nlink == 4) return MZ_SW(4, 0);
  if (P.nlink == 5) return MZ_SW(5, 0);
  return MZ_SW(6, 0);
#undef MZ_SW
}

// ---------------------------------------------------------------- the bit-exact pieces, for CPU tests against the golden vectors
// MazeEnv.get_top_down_view exactly as view_fill_kernel evaluates it (csrc/mz_view.h): float64 entries for one torso / block
// placement, and the in-place fill of fp32 observation rows whose view slots hold the parked block positions
extern "C" void emu_top_down_view(const mz_model* m, double rx, double ry, const double* bxy, double* view) {
  ViewDev V;
  view_dev_from_model(m, &V);
  for (int idx = 0; idx < MZ_VIEW_DIM; idx++) view[idx] = mzv_entry(V, rx, ry, bxy, idx);
}
extern "C" void emu_view_fill_rows(const mz_model* m, int n, int ostride, int view_off, float* rows) {
  ViewDev V;
  view_dev_from_model(m, &V);
  for (int e = 0; e < n; e++) mzv_fill_row(V, rows + (size_t)e * ostride, view_off);
}

extern "C" double emu_hypot(double x, double y) { return mz_hypot(x, y); }

// CollisionDetector.detect + bounce rule on n moves (the code of point_detect_kernel); hit: 0 / 1 bounce / 2 give-up / -1 collinear
extern "C" int emu_point_detect(const mz_model* m, int n, const double* old_xy, const double* new_xy, int32_t* hit, double* point, double* final_xy) {
  PointDev* P = (PointDev*)calloc(1, sizeof(PointDev));
  char err[128];
  int rc = point_dev_from_model(P, m, err, sizeof(err));
  if (rc != MZ_OK) { free(P); return rc; }
  for (int e = 0; e < n; e++) {
    double pt[2];
    hit[e] = point_bounce(*P, old_xy + 2 * e, new_xy + 2 * e, final_xy + 2 * e, pt);
    if (point) { point[2 * e] = pt[0]; point[2 * e + 1] = pt[1]; }
  }
  free(P);
  return MZ_OK;
}

// task_eval_dev on n rows of fp32 observations (obs_dim floats per row; only the first six are read)
extern "C" int emu_task_eval(const mz_model* m, int n, int obs_dim, const float* obs, float* reward, uint8_t* done, int32_t* goal_idx) {
  TaskDev T;
  task_dev_from_model(&T, m);
  for (int e = 0; e < n; e++) {
    float o6[6], r; int tm, gi;
    for (int k = 0; k < 6; k++) o6[k] = k < obs_dim ? obs[(size_t)e * obs_dim + k] : 0.f;
    task_eval_dev(T, o6, &r, &tm, &gi);
    reward[e] = r; done[e] = (uint8_t)tm; goal_idx[e] = gi;
  }
  return MZ_OK;
}

// ---------------------------------------------------------------- generic robot (any tree topology): csrc/generic_dyn.h on one lane
#include "../../mujoco_maze_amd/csrc/generic_dyn.h"

extern "C" int emu_generic_env_step(const mz_model* m, int n, float* qpos, float* qvel, float* warm, int32_t* t, const float* actions, float* obs,
                                    float* reward, uint8_t* done, int32_t* goal_idx, float* info, int32_t* status, char* err, int errlen) {
  GenDev* K = (GenDev*)calloc(1, sizeof(GenDev));
  int rc = gen_dev_from_model(K, m, err, errlen);
  if (rc != MZ_OK) { free(K); return rc; }
  HostCtx cx;
  GenScratch* s = (GenScratch*)calloc(1, sizeof(GenScratch));
  for (int e = 0; e < n; e++) {
    for (int i = 0; i < m->nq; i++) s->qpos[i] = (double)qpos[(size_t)e * m->nq + i];
    for (int i = 0; i < m->nv; i++) { s->qvel[i] = (double)qvel[(size_t)e * m->nv + i]; s->warm[i] = (double)warm[(size_t)e * m->nv + i]; }
    int tt = t[e], gi = -1;
    uint8_t d = 0;
    float r = 0.f, inf[4];
    gen_env_step(cx, *K, *s, actions + (size_t)e * m->nu, obs + (size_t)e * m->obs_dim, &r, &d, &gi, inf, &tt);
    if (m->top_down_view) {  // what mzk_view_fill does on the device: the view from the row's own robot position and the parked block x, y
      ViewDev V;
      view_dev_from_model(m, &V);
      mzv_fill_row(V, obs + (size_t)e * m->obs_dim, m->obs_dim - 1 - MZ_VIEW_DIM);
    }
    for (int i = 0; i < m->nq; i++) qpos[(size_t)e * m->nq + i] = (float)s->qpos[i];
    for (int i = 0; i < m->nv; i++) { qvel[(size_t)e * m->nv + i] = (float)s->qvel[i]; warm[(size_t)e * m->nv + i] = (float)s->warm[i]; }
    t[e] = tt; reward[e] = r; done[e] = d; goal_idx[e] = gi; status[e] = s->status;
    if (info) for (int k = 0; k < 4; k++) info[(size_t)e * 4 + k] = inf[k];
  }
  free(s); free(K);
  return MZ_OK;
}

// narrow-phase routines of the general engine on hand-made poses (tests/test_general_engine.py compares them with the oracle's
// mzo_probe_pair on random poses): kind 0 capsule (size = radius, half length) vs box, 1 box vs box, 2 sphere vs box.
// out: per contact 7 doubles dist | pos | normal (geom1 -> geom2); returns the contact count
extern "C" int emu_probe_pair(int kind, const double* pos1, const double* mat1, const double* size1, const double* pos2, const double* mat2,
                              const double* size2, double margin, int max_con, double* out) {
  int n = 0;
  auto emit = [&](double dist, const double* pos, const double* nrm, const double*) {
    if (n < max_con) { out[7 * n] = dist; for (int k = 0; k < 3; k++) { out[7 * n + 1 + k] = pos[k]; out[7 * n + 4 + k] = nrm[k]; } }
    n++;
  };
  if (kind == 0) {
    const double axis[3] = {mat1[2], mat1[5], mat1[8]};
    gen_capsule_vs_box(pos1, axis, size1[0], size1[1], pos2, mat2, size2, margin, emit);
  } else if (kind == 3) gen_capsule_vs_capsule(pos1, mat1, size1[0], size1[1], pos2, mat2, size2[0], size2[1], margin, emit);
  else if (kind == 1) gen_box_vs_box(pos1, mat1, size1, pos2, mat2, size2, margin, emit);
  else gen_sphere_vs_box(pos1, size1[0], pos2, mat2, size2, margin, emit);
  return n;
}

// one forward-dynamics evaluation of the general engine (float64 state in): qacc [nv], counts = ncon, nlim, Newton iterations, status;
// con (nullable) = per active contact 8 doubles item | dist | pos | normal
extern "C" int emu_generic_forward(const mz_model* m, const double* qpos, const double* qvel, const double* warm, const double* ctrl, double* qacc,
                                   int32_t* counts, double* con, int max_con, char* err, int errlen) {
  GenDev* K = (GenDev*)calloc(1, sizeof(GenDev));
  int rc = gen_dev_from_model(K, m, err, errlen);
  if (rc != MZ_OK) { free(K); return rc; }
  HostCtx cx;
  GenScratch* s = (GenScratch*)calloc(1, sizeof(GenScratch));
  for (int i = 0; i < m->nq; i++) s->qpos[i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { s->qvel[i] = qvel[i]; s->warm[i] = warm ? warm[i] : 0.0; s->fact[i] = 0.0; }
  for (int u = 0; u < m->nu && ctrl; u++) {
    double c = ctrl[u];
    if (m->act_ctrllimited[u]) c = fmin(fmax(c, m->act_ctrlrange[u][0]), m->act_ctrlrange[u][1]);
    s->fact[m->act_dofid[u]] += m->act_gear[u] * (m->act_gainprm[u] * c + m->act_biasprm[u][0]);
  }
  gen_load_topology(cx, *K, *s);
  gen_forward(cx, *K, *s);
  for (int i = 0; i < m->nv; i++) qacc[i] = s->qacc[i];
  counts[0] = s->ncon; counts[1] = s->nlim; counts[2] = s->iters; counts[3] = s->status;
  for (int c = 0; c < s->ncon && c < max_con && con; c++) {
    con[8 * c] = s->citem[c]; con[8 * c + 1] = s->cdist[c];
    for (int k = 0; k < 3; k++) { con[8 * c + 2 + k] = s->cpos[c][k]; con[8 * c + 5 + k] = s->cnrm[c][k]; }
  }
  free(s); free(K);
  return MZ_OK;
}

// raw mj_step x nsteps of the general engine on a float64 state (no maze logic, no fp32 rounding), for stage-level comparisons
// with the oracle's mzo_raw_steps
extern "C" int emu_generic_raw_steps(const mz_model* m, double* qpos, double* qvel, double* warm, const double* ctrl, int nsteps, int32_t* status,
                                     char* err, int errlen) {
  GenDev* K = (GenDev*)calloc(1, sizeof(GenDev));
  int rc = gen_dev_from_model(K, m, err, errlen);
  if (rc != MZ_OK) { free(K); return rc; }
  HostCtx cx;
  GenScratch* s = (GenScratch*)calloc(1, sizeof(GenScratch));
  for (int i = 0; i < m->nq; i++) s->qpos[i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { s->qvel[i] = qvel[i]; s->warm[i] = warm ? warm[i] : 0.0; s->fact[i] = 0.0; }
  for (int u = 0; u < m->nu && ctrl; u++) {
    double c = ctrl[u];
    if (m->act_ctrllimited[u]) c = fmin(fmax(c, m->act_ctrlrange[u][0]), m->act_ctrlrange[u][1]);
    s->fact[m->act_dofid[u]] += m->act_gear[u] * (m->act_gainprm[u] * c + m->act_biasprm[u][0]);
  }
  gen_load_topology(cx, *K, *s);
  for (int k = 0; k < nsteps; k++) gen_mj_step(cx, *K, *s);
  for (int i = 0; i < m->nq; i++) qpos[i] = s->qpos[i];
  for (int i = 0; i < m->nv; i++) { qvel[i] = s->qvel[i]; if (warm) warm[i] = s->warm[i]; }
  if (status) *status = s->status;
  free(s); free(K);
  return MZ_OK;
}
