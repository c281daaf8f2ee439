"""EINV2 networks — mirror of the reference's `models/einv2.py` registry module. Filled in by einv2 support."""
from . import accdoa

HTSAT = HTSAT_SEDDOA = CRNN = ConvConformer = PASST = accdoa._NotBuilt
