// mz_view.h — MazeEnv.get_top_down_view (maze_env.py:262-349) for tasks with TOP_DOWN_VIEW (maze_env.py:54,351-369).
//
// The view is a robot-centred 5 x 5 grid with three channels (walls, chasms, movable blocks).  Every source — a BLOCK or
// CHASM cell of the maze, or a movable block's body position — is a unit square in grid coordinates
//     row = 2 + (y - robot_y + scale/2) / scale,   col = 2 + (x - robot_x + scale/2) / scale        (maze_env.py:90-93)
// that the reference splats over the cell (int(row), int(col)) and its eight neighbours with the overlap areas as weights
// (`update_view`, maze_env.py:268-320).  The weights factor into a row weight times a column weight, a target cell
// receives at most one term from each source, and the reference adds the sources in a fixed order (cells row-major, then
// the blocks in creation order) — so entry (R, C, d) is a sum, in that order, of  w(row_k, R) * w(col_k, C)  over the sources
// of channel d: a gather that one thread can evaluate per entry in the reference's own float64 arithmetic.
//
// Reference quirk kept: the cell index is Python's int() (truncation towards zero) while the fraction is Python's `% 1`
// (floor modulo), so a source at row -0.3 lands on cell 0 with fraction 0.7.
//
// The step / reset kernels write the observation rows with the view's 75 entries left open (the movable blocks' x, y are
// parked in the first entries); `view_fill_rows` then fills them from the row's own robot position.  Host-callable too
// (tests/emu pins it against the reference's views, tests/golden/views.json).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/mazestep.h"

#if defined(__HIPCC__)
#define MZV_HD __host__ __device__ inline
#else
#define MZV_HD inline
#endif

struct ViewDev {
  int on, rows, cols, nblock;
  uint32_t wall[MZ_MAX_GRID], chasm[MZ_MAX_GRID];  // bit j of word i: cell (i, j) is a BLOCK / a CHASM
  double scale, tx, ty;                             // maze_size_scaling, _init_torso_x / _y
};

static inline void view_dev_from_model(const mz_model* m, ViewDev* v) {
  v->on = m->top_down_view;
  v->rows = m->grid_rows;
  v->cols = m->grid_cols;
  v->nblock = m->nblock;
  v->scale = m->maze_scale;
  v->tx = m->torso_x;
  v->ty = m->torso_y;
  for (int i = 0; i < MZ_MAX_GRID; i++) {
    v->wall[i] = v->chasm[i] = 0u;
    for (int j = 0; j < MZ_MAX_GRID; j++)
      if (i < m->grid_rows && j < m->grid_cols) {
        if (m->grid[i][j] == MZ_CELL_BLOCK) v->wall[i] |= 1u << j;
        if (m->grid[i][j] == MZ_CELL_CHASM) v->chasm[i] |= 1u << j;
      }
  }
}

// Every function below opens with `#pragma clang fp contract(off) reciprocal(off) reassociate(off)`: the reference's float64
// arithmetic, operation by operation, whatever the flags of the including translation unit.

// Python's `x % 1` for floats (CPython float_rem): fmod, moved into [0, 1)
MZV_HD double mzv_mod1(double x) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  double r = fmod(x, 1.0);
  if (r != 0.0) {
    if (r < 0.0) r += 1.0;
  } else {
    r = 0.0;
  }
  return r;
}

// weight a source at fractional grid coordinate `rho` gives to the target index R (maze_env.py:283-320, one axis)
MZV_HD double mzv_weight(double rho, int R) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  if (!(fabs(rho) < 1e9)) return 0.0;
  const int base = (int)rho;
  const double f = mzv_mod1(rho);
  if (R == base) return fmin(1.0, f + 0.5) - fmax(0.0, f - 0.5);
  if (R == base - 1) return fmax(0.0, 0.5 - f);
  if (R == base + 1) return fmax(0.0, f - 0.5);
  return 0.0;
}

// entry idx = (R * 5 + C) * 3 + d of the flattened view for a torso at (rx, ry) and movable blocks at bxy[2 k], bxy[2 k + 1]
MZV_HD double mzv_entry(const ViewDev& V, double rx, double ry, const double* bxy, int idx) {
#pragma clang fp contract(off) reciprocal(off) reassociate(off)
  const int d = idx % 3, C = (idx / 3) % 5, R = idx / 15;
  const double sc = V.scale;
  double acc = 0.0;
  if (d < 2) {
    for (int i = 0; i < V.rows; i++) {
      const uint32_t bits = d == 0 ? V.wall[i] : V.chasm[i];
      if (!bits) continue;
      double y = (double)i * sc - V.ty;
      y = y - ry;
      const double wr = mzv_weight(2.0 + (y + sc / 2.0) / sc, R);
      if (wr == 0.0) continue;  // (a zero term leaves the sum as it is)
      for (int j = 0; j < V.cols; j++)
        if (bits >> j & 1u) {
          double x = (double)j * sc - V.tx;
          x = x - rx;
          acc += wr * mzv_weight(2.0 + (x + sc / 2.0) / sc, C);
        }
    }
  } else {
    for (int b = 0; b < V.nblock; b++) {
      const double x = bxy[2 * b] - rx, y = bxy[2 * b + 1] - ry;
      acc += mzv_weight(2.0 + (y + sc / 2.0) / sc, R) * mzv_weight(2.0 + (x + sc / 2.0) / sc, C);
    }
  }
  return acc;
}

// Fill the view entries of one observation row: robot position = row[0:2] (every robot's observation starts with the torso's
// x, y), block positions = the values the step / reset kernel parked at row[view_off ...].
// Precision: the arithmetic is the reference's float64 arithmetic, but on the DEVICE its inputs are the fp32-rounded positions
// of the observation row, not the float64 state — the device view differs from the float64 oracle's by ~1e-7 per entry and can
// pick the other cell for a position within fp32 round-off of a cell boundary.  Bit-for-bit parity with the reference's views
// (tests/golden/views.json) holds for the host / emulation path, which is handed float64 positions.
MZV_HD void mzv_fill_row(const ViewDev& V, float* row, int view_off) {
  const double rx = (double)row[0], ry = (double)row[1];
  double bxy[8];
  for (int k = 0; k < 2 * V.nblock && k < 8; k++) bxy[k] = (double)row[view_off + k];
  for (int idx = 0; idx < MZ_VIEW_DIM; idx++) row[view_off + idx] = (float)mzv_entry(V, rx, ry, bxy, idx);
}
