// Internal interface of the tiled form of nvl_linear_wide (gemm_tile.hip), used by gemm_wide.hip's planner / launcher.
#pragma once
#include <stdint.h>
// Is the shape one the tiled form takes (33 ... 256 rows, k a multiple of 64, whole 16-column tiles)?
bool nvl_tile_covers(int64_t m, int n, int k, int mode);
// Workgroups along N (128 weight rows each; 64 output columns with the SiLU epilogue).
int nvl_tile_workgroups(int n, int mode);
// Enqueue. mode 0 / 1 need split == 1 (out = bf16); mode 2 writes fp32 slabs [split][m][n]. Tile-packed weights only.
int nvl_tile_launch(const void* x, const void* w_packed, void* out, int64_t m, int n, int k, int mode, int split, void* stream);
