"""occm_amd -- MI355X (gfx950) native implementation of the occm data-parallel training hot path.

Host side mirrors the reference's Python surface (``models.sslassist.AModel``, ``models.xlsr.SSLModel``,
``models.senet``, ``losses.custom_loss``, ``RawBoost``, ``data_utils_SSL``, ``oc_training``,
``oc_classifier``, ``evaluate_metrics``, ``calculate_eer``); the arithmetic runs in hand-written HIP
kernels behind the C ABI of ``libocc_hip.so`` (``include/occ_hip.h``).  There is no CPU fallback: a
missing library or a missing GPU raises.
"""
__version__ = "0.1.0"
