"""data/split_dataset_tiledpred.py of the reference: a dataset whose index is a tile of the ShiftBoundary tiling.

``SplitDatasetTiledPred(data_type, data_location, patch_size, grid_size=..., **SplitDataset kwargs)`` is the
reference's signature (frames resident on the GPU, batches of normalised tiles cut by one HIP launch:
``split_dataset.py`` here).  The in-memory form ``SplitDatasetTiledPred(frames, patch_size, grid_size)`` of round 1
((N,H,W,2) array of already normalised channels, numpy items) is kept."""
import numpy as np

from .tiling_manager import TileIndexManager, TilingMode


class _FramesTiledPred:
    """``frames``: (N,H,W,2) array of the two (normalised) channels.  Item i ->
    {'input': (1,p,p) = ch0+ch1 weighted sum, 'target': (2,p,p)} like SplitDataset.__getitem__."""

    def __init__(self, frames, patch_size, grid_size=None, channel_weights=(1, 1)):
        self._frames = np.asarray(frames, dtype=np.float32)
        self._patch_size = patch_size
        if grid_size is None:
            grid_size = patch_size // 2
        n, h, w, _ = self._frames.shape
        self._w = channel_weights
        self.tile_manager = TileIndexManager((n, h, w), (1, grid_size, grid_size),
                                             (1, patch_size, patch_size), TilingMode.ShiftBoundary)

    def __len__(self):
        return self.tile_manager.total_grid_count()

    def patch_location(self, index):
        return self.tile_manager.get_patch_location_from_dataset_idx(index)

    def __getitem__(self, index):
        n, y, x = self.patch_location(index)
        p = self._patch_size
        target = np.moveaxis(self._frames[n, y:y + p, x:x + p, :], -1, 0).copy()
        inp = (self._w[0] * target[0:1] + self._w[1] * target[1:2]).astype(np.float32)
        return {"input": inp, "target": target}


class SplitDatasetTiledPred:
    def __new__(cls, *args, **kwargs):
        if args and isinstance(args[0], np.ndarray):
            return _FramesTiledPred(*args, **kwargs)
        from .split_dataset import SplitDatasetTiledPred as _Device
        return _Device(*args, **kwargs)
