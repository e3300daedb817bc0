// mz_device.h — device-side pieces shared by the kernel translation units (ant_kernels.hip, planar_kernels.hip):
// the lane-group context, the counter-based RNG of the reset distribution, episode seeding.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Index of base-layout observation element i (0 .. base_dim - 1: the observation without a top-down view, whose last element
// is t * 0.001) in a row of `ostride` floats: with a view (ostride = base_dim + MZ_VIEW_DIM) the time entry moves behind it
// (maze_env.py:369); the view's own entries are filled by mzk_view_fill, which finds the movable blocks' x, y parked at
// row[base_dim - 1 ...].
__host__ __device__ inline int obs_slot(int i, int base_dim, int ostride) { return i == base_dim - 1 ? ostride - 1 : i; }

// Workgroup -> block of environments, XCD-aware.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with an L2 of
// its own; with block b = workgroup id, the 16 one-wave workgroups whose envs share a 128-byte line of an SoA state array
// (32 envs x 4 B; the Point / chain kernels) sit on 8 different XCDs and the line is fetched — and partially written — by all
// of them (PointUMaze, round 3: FETCH + WRITE = 3.9 x the algorithmic bytes).  Here the workgroups of XCD x (ids = x mod 8) take
// the x-th contiguous eighth of the blocks, so a line has one home.  Placement is a speed hint only (the id -> XCD map is not
// architectural): any map is correct, every block is taken exactly once.
__device__ __forceinline__ int xcd_block(int wg, int nwg) {
#ifdef MZ_EXP_NOXCD  // A/B experiment build (tools/exp_build.sh NOXCD): the identity map
  return wg;
#endif
  const int per = nwg >> 3;                       // blocks per XCD (the remainder nwg & 7 keeps the identity map)
  return wg < 8 * per ? (wg & 7) * per + (wg >> 3) : wg;
}

// ------------------------------------------------------------------ device context of a lane group
template <int G, bool PROF = false>
struct DevCtx {
  static constexpr int nlanes = G;
  static constexpr bool row_solver = G >= 16;  // plain ant: Newton iteration resident in one 16-lane DPP row (ant_newton_rows.h)
  static constexpr int NLC = 32;
  int l;
  float lc[NLC] = {};  // per-lane constants of the plain ant's quad layout (ant_forward_rows.h ant_lane_consts); unused elsewhere
  bool mfma = false;   // the row solver's Hessian fold on the matrix cores (ant_newton_rows.h fold_h_mfma): set by the kernel, a compile-time constant after inlining
  // phase timer (PROF builds only): lane 0 of the group accumulates shader cycles since the previous tick
  template <class S>
  __device__ __forceinline__ void tick(S& s, int id) const {
    if constexpr (PROF) {
      if (l == 0) {
        unsigned long long now = __builtin_amdgcn_s_memtime();
        s.prof[id] += (unsigned)(now - s.prof_t0);
        s.prof_t0 = now;
      }
    }
  }
  __device__ __forceinline__ int lane0() const { return l; }
  // A lane group never spans wavefronts and LDS operations of one wavefront execute in order, so a
  // hand-off between lanes of a group needs no s_barrier: a wavefront-scope fence (no instruction, it only
  // pins the compiler's ordering of the LDS stores before and loads after) is a complete phase boundary.
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // All-reduce (sum) inside the lane group.  Rows of 16 lanes reduce with four DPP moves (quad_perm xor 1,
  // xor 2, row_half_mirror, row_mirror: VALU-rate, no LDS crossbar).
  // (contract(off): partners of a butterfly step must add the same two numbers — x + y, never fma(p, q, y) with x = p * q still
  // open — or the lanes of a group end up with sums that differ in the last bit and part ways at the next branch:
  // ant_newton_rows.h rsum)
  static __device__ __forceinline__ float dpp_add(float x, const int ctrl_sel) {
#pragma clang fp contract(off)
    int xi = __float_as_int(x), yi;
    switch (ctrl_sel) {
      case 0: yi = __builtin_amdgcn_mov_dpp(xi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
      case 1: yi = __builtin_amdgcn_mov_dpp(xi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
      case 2: yi = __builtin_amdgcn_mov_dpp(xi, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
      default: yi = __builtin_amdgcn_mov_dpp(xi, 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return x + __int_as_float(yi);
  }
  // Cross-row steps stay on the VALU as well (no ds_bpermute round trip through the LDS crossbar, ~100 cycles each on the
  // serial critical path of the line search): row_bcast:15 adds lane 15 of rows 0 / 2 into every lane of rows 1 / 3,
  // row_bcast:31 adds lane 31 into rows 2 and 3; the group total then sits in the group's last row and is handed back to
  // all lanes with one v_readlane per group (scalar registers) and a select.
  __device__ __forceinline__ float gsum(float x) const {
#pragma clang fp contract(off)
    if constexpr (G >= 2) x = dpp_add(x, 0);
    if constexpr (G >= 4) x = dpp_add(x, 1);
    if constexpr (G >= 8) x = dpp_add(x, 2);
    if constexpr (G >= 16) x = dpp_add(x, 3);
    if constexpr (G == 32) {
      x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xA, 0xF, false));  // rows 1, 3 += lane 15 of rows 0, 2
      const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
      const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
      x = __lane_id() < 32 ? t0 : t1;
    }
    if constexpr (G >= 64) {
      x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xA, 0xF, false));
      x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x143, 0xC, 0xF, false));  // rows 2, 3 += lane 31
      x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    }
    return x;
  }
  // fp64 paths (Point): the same reduction on the two halves of a double — DPP moves instead of `ds_bpermute` round trips
  // (a butterfly of shuffles cost ~8 LDS-crossbar latencies per sum on the serial path of the Point's Newton iteration)
  template <int CTRL, int ROW_MASK, bool BOUND>
  static __device__ __forceinline__ double dpp_movd(double x) {
    const long long b = __double_as_longlong(x);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    if constexpr (BOUND) { lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xF, true); }
    else { lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false); }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
  static __device__ __forceinline__ double readlaned(double x, int lane) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
  __device__ __forceinline__ double gsum(double x) const {
#pragma clang fp contract(off)
    if constexpr (G >= 2) x += dpp_movd<0xB1, 0xF, true>(x);
    if constexpr (G >= 4) x += dpp_movd<0x4E, 0xF, true>(x);
    if constexpr (G >= 8) x += dpp_movd<0x141, 0xF, true>(x);
    if constexpr (G >= 16) x += dpp_movd<0x140, 0xF, true>(x);
    if constexpr (G == 32) {
      x += dpp_movd<0x142, 0xA, false>(x);  // rows 1, 3 += lane 15 of rows 0, 2 (other rows add zero)
      const double t0 = readlaned(x, 16), t1 = readlaned(x, 48);
      x = __lane_id() < 32 ? t0 : t1;
    }
    if constexpr (G >= 64) {
      x += dpp_movd<0x142, 0xA, false>(x);
      x += dpp_movd<0x143, 0xC, false>(x);  // rows 2, 3 += lane 31
      x = readlaned(x, 63);
    }
    return x;
  }
  // sum over the 16-lane DPP row of the lane (the whole group when it is narrower): four DPP steps, no cross-row traffic.  The rows
  // of a wider group reduce independently (point_bare.h: every row of a group holds the same contacts and gets the same bits).
  __device__ __forceinline__ double rowsum(double x) const {
#pragma clang fp contract(off)
    if constexpr (G >= 2) x += dpp_movd<0xB1, 0xF, true>(x);
    if constexpr (G >= 4) x += dpp_movd<0x4E, 0xF, true>(x);
    if constexpr (G >= 8) x += dpp_movd<0x141, 0xF, true>(x);
    if constexpr (G >= 16) x += dpp_movd<0x140, 0xF, true>(x);
    return x;
  }
  __device__ __forceinline__ bool any(bool p) const { return __any(p) != 0; }
  // ballot restricted to this lane group, bit k = lane k of the group
  __device__ __forceinline__ unsigned long long gballot(bool p) const {
    const unsigned long long b = __ballot(p);
    if constexpr (G >= 64) return b;
    else return (b >> (__lane_id() - (unsigned)l)) & ((1ULL << (G & 63)) - 1ULL);
  }
  // any() restricted to this lane group: one ballot, no shuffles
  __device__ __forceinline__ bool gany(bool p) const {
    unsigned long long b = __ballot(p);
    unsigned lane = __lane_id();
    unsigned long long gm = (G >= 64) ? ~0ULL : (((1ULL << (G & 63)) - 1ULL) << (lane - (unsigned)l));
    return (b & gm) != 0ULL;
  }
};

// ------------------------------------------------------------------ RNG (same definition as the oracle's mzo_rng_u32)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t env, uint32_t counter) {
  uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (env + 1));
  z = mix64(z + 0x9E3779B97F4A7C15ULL * ((uint64_t)counter + 1));
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float rng_u01(uint64_t seed, uint64_t env, uint32_t c) { return (float)(rng_u32(seed, env, c) >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float rng_normal(uint64_t seed, uint64_t env, uint32_t c) {
  float u1 = ((float)(rng_u32(seed, env, c) >> 8) + 1.0f) * (1.0f / 16777216.0f);
  float u2 = rng_u01(seed, env, c + 1);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
// reset distribution (ant.py:84-96, point.py:71-81): qpos0 + U(-.1,.1); qvel by kind
__device__ __forceinline__ float reset_qpos(float q0, uint64_t seed, uint64_t env, int i) { return q0 + (-0.1f + 0.2f * rng_u01(seed, env, (uint32_t)i)); }
__device__ __forceinline__ float reset_qvel(int kind, int nq, uint64_t seed, uint64_t env, int i) {
  uint32_t c = (uint32_t)(nq + 2 * i);
  if (kind == 0) return 0.1f * rng_normal(seed, env, c);
  if (kind == 1) return 0.1f * rng_u01(seed, env, c);
  return -0.1f + 0.2f * rng_u01(seed, env, c);
}
__host__ __device__ __forceinline__ uint64_t episode_seed(uint64_t seed, uint32_t episode) {
  return episode == 0 ? seed : mix64(seed ^ (0xD6E8FEB86659FD93ULL * (uint64_t)episode));
}
