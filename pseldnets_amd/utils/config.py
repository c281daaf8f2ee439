"""Mirror of the reference's feature-extractor selection hook (reference: utils/config.py:24-32)."""
from . import feature


def get_afextractor(cfg):
    """Get audio feature extractor: 'logmelIV' | 'logmel' -> module, anything else -> None."""
    kind = cfg['data']['audio_feature']
    if kind == 'logmelIV':
        return feature.LogmelIV_Extractor(cfg)
    if kind == 'logmel':
        return feature.Logmel_Extractor(cfg)
    return None
