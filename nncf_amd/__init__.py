"""nncf_amd -- MI355X-native implementation of the NNCF sampling + scoring hot path.

Host side mirrors the reference's Python interfaces for this path (sampler.MultinomialSampler,
model_framework.get_model, models/train_*.Trainer); compute runs in hand-written HIP
kernels for gfx950 behind the C ABI in include/nncf_hip.h (libnncf_hip.so).
"""
__version__ = "0.1.0"
