from . import accdoa, einv2, multi_accdoa  # noqa: F401  (registry modules, as reference models/__init__.py:1-3)
