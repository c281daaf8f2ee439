"""pseldnets_amd — MI355X-native hot path of PSELDNets (features -> HTS-AT -> ACCDOA heads -> loss -> step).

Host side mirrors the reference's Python seams (feature extractor, network registry, loss); all arithmetic runs
in hand-written gfx950 kernels behind the C ABI in include/pseld_hip.h.
"""
__version__ = "0.1.0"
