"""mofo_amd -- MI355X-native (gfx950) implementation of the MOFO / VideoMAE masked-video-autoencoder PRETRAINING step.

Module names mirror the reference files they stand in for:
    modeling_pretrain, engine_for_pretraining, masking_generator, optim_factory, utils
Compute runs in hand-written HIP kernels behind the C-ABI in include/mofo_hip.h (mofo_amd/libmofo_hip.so);
there is no CPU fallback.
"""
__version__ = "0.1.0"
