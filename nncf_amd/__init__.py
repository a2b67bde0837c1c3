"""nncf_amd -- MI355X-native implementation of the NNCF sampling + scoring hot path.

Host side mirrors the reference's Python interfaces for this path (sampler.MultinomialSampler,
model_framework.get_model, models/train_*.Trainer); compute runs in hand-written HIP
kernels for gfx950 behind the C ABI in include/nncf_hip.h (libnncf_hip.so).
"""
__version__ = "0.1.0"

import os as _os

# Kernel arguments in device memory (the ROCm default on this image): with HIP_FORCE_DEV_KERNARG=0 every launch of this
# path -- chains of dependent 5-20 us kernels with 300-700 B of arguments -- fetches them from host memory at dispatch and the
# headline step goes from 45.7 to 61.0 us (measured, profiles/r03_mfma_peak.txt).  Only a default: an explicit setting wins,
# and it has to be in the environment before the HIP runtime initialises (importing this package before the first GPU call).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
