"""Multi-ACCDOA networks on MI355X — mirror of the reference's `models/multi_accdoa.py` (CRNN :5-16, ConvConformer :18-27, HTSAT :29-44, PASST :46-54):
the ACCDOA net with a 3 tracks x 3 axes x C head and output key 'multi_accdoa'."""
from . import accdoa


class HTSAT(accdoa.HTSAT):
    out_key = 'multi_accdoa'
    tracks_axes = 9


class PASST(accdoa.PASST):
    out_key = 'multi_accdoa'
    tracks_axes = 9


class CRNN(accdoa.CRNN):
    out_key = 'multi_accdoa'
    tracks_axes = 9


class ConvConformer(accdoa.ConvConformer):
    out_key = 'multi_accdoa'
    tracks_axes = 9
