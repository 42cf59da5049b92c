"""Drop-in for the reference's ``masking_generator.py``: same class names, constructor arguments, ``__call__``
results (float64 0/1 vectors, 1 = masked) and -- deliberately -- the same use of numpy's GLOBAL RNG, so that a seeded
run draws bit-identical masks.  Host-side integer work (negligible cost, runs in DataLoader workers)."""
import numpy as np


class TubeMaskingGenerator:
    """masking_generator.py:3-24: one shuffled per-frame pattern repeated over all temporal slots (a "tube")."""

    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = input_size
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.frames * self.num_patches_per_frame
        self.num_masks_per_frame = int(mask_ratio * self.num_patches_per_frame)
        self.total_masks = self.frames * self.num_masks_per_frame

    def __repr__(self):
        return "Maks: total patches {}, mask patches {}".format(self.total_patches, self.total_masks)

    def __call__(self):
        pattern = np.concatenate([np.zeros(self.num_patches_per_frame - self.num_masks_per_frame),
                                  np.ones(self.num_masks_per_frame)])
        np.random.shuffle(pattern)
        return np.tile(pattern, (self.frames, 1)).reshape(-1)


class TubeMaskingGenerator_BB:
    """masking_generator.py:27-85 (MOFO motion-bounding-box mask).  Reference behaviour kept on purpose:
    only ``bb[0]`` is consulted; the box x-range is tested against the patch ROW and the y-range against the COLUMN;
    a patch is "in the box" unless it misses on BOTH axes; ``mask_ratio_BB`` of the shuffled in-box patches (capped at
    the per-frame budget) are forced masked; the rest of the budget is drawn from patches 0..num_masks_per_frame-1 only."""

    def __init__(self, input_size, mask_ratio, mask_ratio_BB):
        self.frames, self.height, self.width = input_size
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.frames * self.num_patches_per_frame
        self.num_masks_per_frame = int(mask_ratio * self.num_patches_per_frame)
        self.total_masks = self.frames * self.num_masks_per_frame
        self.mask_ratio = mask_ratio
        self.mask_ratio_BB = mask_ratio_BB

    def __repr__(self):
        return "Maks: total patches {}, mask patches {}".format(self.total_patches, self.total_masks)

    def __call__(self, bb):
        x1, y1, x2, y2 = bb[0][0], bb[0][1], bb[0][2], bb[0][3]
        in_box = []
        for row in range(self.height):
            misses_x = (x1 > 16 * row + 16) or (x2 < 16 * row)
            for col in range(self.width):
                misses_y = (y1 > 16 * col + 16) or (y2 < 16 * col)
                if not (misses_x and misses_y):
                    in_box.append(row * self.width + col)
        np.random.shuffle(in_box)
        forced = in_box[:min(self.num_masks_per_frame, int(len(in_box) * self.mask_ratio_BB))]
        pattern = np.zeros(self.num_patches_per_frame)
        pattern[forced] = 1
        rest = np.setdiff1d(np.arange(self.num_masks_per_frame), forced)
        np.random.shuffle(rest)
        pattern[rest[:self.num_masks_per_frame - len(forced)]] = 1
        return np.tile(pattern, (self.frames, 1)).reshape(-1)
