"""Drop-in for the reference's ``masking_generator.py``: same class names, constructor arguments, attributes, ``repr`` and
``__call__`` results (float64 0/1 vectors, 1 = masked) and -- deliberately -- the same use of numpy's GLOBAL RNG, so that
a seeded run draws bit-identical masks.  Host-side integer work (negligible cost, runs in DataLoader workers)."""
import numpy as np


class _TubeGeometry:
    """token grid (frames, height, width) and the per-frame masking budget shared by both generators"""

    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = input_size
        per_frame = self.height * self.width
        self.num_patches_per_frame = per_frame
        self.num_masks_per_frame = int(mask_ratio * per_frame)
        self.total_patches = self.frames * per_frame
        self.total_masks = self.frames * self.num_masks_per_frame

    def __repr__(self):   # the reference's spelling
        return f"Maks: total patches {self.total_patches}, mask patches {self.total_masks}"

    def _tube(self, frame_pattern):
        """the same per-frame pattern in every temporal slot"""
        return np.tile(frame_pattern, (self.frames, 1)).reshape(-1)


class TubeMaskingGenerator(_TubeGeometry):
    """masking_generator.py:3-24: one shuffled per-frame pattern repeated over all temporal slots (a "tube")."""

    def __call__(self):
        n_keep = self.num_patches_per_frame - self.num_masks_per_frame
        pattern = np.concatenate([np.zeros(n_keep), np.ones(self.num_masks_per_frame)])
        np.random.shuffle(pattern)
        return self._tube(pattern)


class TubeMaskingGenerator_BB(_TubeGeometry):
    """masking_generator.py:27-85 (MOFO motion-bounding-box mask).  Reference behaviour kept on purpose:
    only ``bb[0]`` is consulted; the box x-range is tested against the patch ROW and the y-range against the COLUMN;
    a patch is "in the box" unless it misses on BOTH axes; ``mask_ratio_BB`` of the shuffled in-box patches (capped at
    the per-frame budget) are forced masked; the rest of the budget is drawn from patches 0..num_masks_per_frame-1 only."""

    def __init__(self, input_size, mask_ratio, mask_ratio_BB):
        super().__init__(input_size, mask_ratio)
        self.mask_ratio, self.mask_ratio_BB = mask_ratio, mask_ratio_BB

    def _in_box(self, box):
        x1, y1, x2, y2 = (box[i] for i in range(4))
        hit = []
        for row in range(self.height):
            misses_x = (x1 > 16 * row + 16) or (x2 < 16 * row)
            for col in range(self.width):
                misses_y = (y1 > 16 * col + 16) or (y2 < 16 * col)
                if not (misses_x and misses_y):
                    hit.append(row * self.width + col)
        return hit

    def __call__(self, bb):
        budget = self.num_masks_per_frame
        in_box = self._in_box(bb[0])
        np.random.shuffle(in_box)
        forced = in_box[:min(budget, int(len(in_box) * self.mask_ratio_BB))]
        pattern = np.zeros(self.num_patches_per_frame)
        pattern[forced] = 1
        fill = np.setdiff1d(np.arange(budget), forced)
        np.random.shuffle(fill)
        pattern[fill[:budget - len(forced)]] = 1
        return self._tube(pattern)


class DeviceTubeMaskingGenerator(_TubeGeometry):
    """Tube masks drawn ON the device for a whole batch (no mask crosses PCIe; not in the reference).
    Same distribution as TubeMaskingGenerator -- per clip one pattern with exactly ``num_masks_per_frame`` masked patches per
    frame, repeated over the temporal slots, every subset equally likely -- but its OWN random stream (a counter-based integer
    mixer keyed by ``seed`` and a running clip counter; ``oracle.pretrain_oracle.device_tube_masks`` restates it): the
    reference's masks come from numpy's global Mersenne-Twister inside DataLoader workers, which a kernel cannot continue, so
    runs that must reproduce the reference's masks keep the host class above.

        gen = DeviceTubeMaskingGenerator((8, 14, 14), 0.9, seed=0)
        mask = gen(batch_size, out=model.input_buffers(batch_size, n_vis)[1])     # uint8 [B, 1568] on the device, 1 = masked
    """

    def __init__(self, input_size, mask_ratio, seed=0, rank=None, world_size=None):
        """``rank`` / ``world_size`` (data parallelism): the clip counter of a draw is ``step * global_batch + rank * batch_size + i``,
        so ranks that share one ``seed`` still draw DIFFERENT masks (the reference's workers are seeded ``seed + rank``,
        run_mae_pretraining.py:166) and a job's masks do not depend on how its global batch is split over ranks.  Left at None
        they come from the initialised torch.distributed group, else from RANK / WORLD_SIZE (utils.py:277-296), else 0 / 1."""
        super().__init__(input_size, mask_ratio)
        self.seed, self.clips_drawn = int(seed), 0
        if rank is None or world_size is None:
            r, w = self._ambient_rank()
            rank = r if rank is None else rank
            world_size = w if world_size is None else world_size
        self.rank, self.world_size = int(rank), max(1, int(world_size))

    @staticmethod
    def _ambient_rank():
        import os
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_rank(), dist.get_world_size()
        except Exception:
            pass
        return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))

    def state_dict(self):
        """what a checkpoint needs so that a resumed run continues the mask stream instead of replaying step 0"""
        return {"seed": self.seed, "clips_drawn": self.clips_drawn, "world_size": self.world_size}

    def load_state_dict(self, state):
        """``clips_drawn`` counts PER RANK: the global index of a step's first clip is clips_drawn * world_size (+ rank * batch), so a
        resume under another world size continues at a different place of the stream -- said aloud, not silently"""
        self.seed, self.clips_drawn = int(state["seed"]), int(state["clips_drawn"])
        saved = int(state.get("world_size", self.world_size))
        if saved != self.world_size:
            import warnings
            warnings.warn(f"device mask generator: checkpoint written with world_size {saved}, resuming with {self.world_size}: the "
                          f"mask stream continues at global clip {self.clips_drawn * self.world_size} instead of {self.clips_drawn * saved}")

    def __call__(self, batch_size, out=None, device=None):
        import torch
        from . import ops
        if out is None:
            out = torch.empty(batch_size, self.total_patches, dtype=torch.uint8, device=device or "cuda")
        counter = self.clips_drawn * self.world_size + self.rank * batch_size     # global index of this rank's first clip of the step
        ops.tube_masks(self.seed, counter, self.frames, self.num_patches_per_frame, self.num_masks_per_frame, out)
        self.clips_drawn += batch_size
        return out
