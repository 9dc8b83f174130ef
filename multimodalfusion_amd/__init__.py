"""multimodalfusion_amd -- MI355X-native attention-MIL + multimodal-fusion hot path.

Drop-in for the reference's `models.*` / `utils.loss_utils` surface (same class names,
constructor signatures, forward(**kwargs) contract and state_dict keys); the arithmetic runs
in hand-written HIP kernels (csrc/) behind the C ABI of include/mmf_amil.h.
"""
__version__ = "0.1.0"
