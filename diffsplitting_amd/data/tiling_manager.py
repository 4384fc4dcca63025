"""data/tiling_manager.py of the reference: ``TilingMode`` and ``TileIndexManager``,
backed by the library's tile planner (dsx_tile_plan)."""
from dataclasses import dataclass

import numpy as np

from .tiling import PAD, SHIFT, TRIM, TilePlan


class TilingMode:
    TrimBoundary = TRIM
    PadBoundary = PAD
    ShiftBoundary = SHIFT


@dataclass
class TileIndexManager:
    data_shape: tuple
    grid_shape: tuple
    patch_shape: tuple
    tiling_mode: int

    def __post_init__(self):
        self._plan = TilePlan(self.data_shape, self.grid_shape, self.patch_shape, self.tiling_mode)

    @property
    def plan(self):
        return self._plan

    def patch_offset(self):
        return (np.array(self.patch_shape) - np.array(self.grid_shape)) // 2

    def total_grid_count(self):
        return self._plan.total

    def get_location_from_dataset_idx(self, dataset_idx):
        return tuple(int(v) for v in self._plan.grid_start[dataset_idx])

    def get_patch_location_from_dataset_idx(self, dataset_idx):
        return tuple(int(v) for v in self._plan.patch_start[dataset_idx])
