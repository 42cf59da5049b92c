"""mofo_amd -- MI355X-native (gfx950) implementation of the MOFO / VideoMAE masked-video-autoencoder PRETRAINING step.

Module names mirror the reference files they stand in for:
    modeling_pretrain, engine_for_pretraining, masking_generator, optim_factory, utils
Compute runs in hand-written HIP kernels behind the C-ABI in include/mofo_hip.h (mofo_amd/libmofo_hip.so);
there is no CPU fallback.
"""
import os as _os

# more hardware queues than HIP's default of 4, so that the weight-gradient side stream does not share one with the compute stream
# once torch.distributed adds its own streams (bench.py: 0.24 ms per data-parallel step); only effective if this import comes
# before the first HIP call of the process -- launchers set it themselves at their top
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
