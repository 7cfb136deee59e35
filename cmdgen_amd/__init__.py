"""cmdgen_amd - MI355X-native DiffPhar denoising loop (pocket-conditioned EGNN DDPM sampler).

Host side mirrors the reference's Python interface for this path
(``PharPocketDDPM``, ``ConditionalDDPM``, ``EGNNDynamics``, ``generate_phars``);
all arithmetic on the path runs in hand-written HIP kernels for gfx950 behind
the C ABI declared in ``include/cmdgen_hip.h`` (``csrc/``).  There is no CPU
fallback: using the model without the built library raises.
"""
__version__ = '0.1.0'
