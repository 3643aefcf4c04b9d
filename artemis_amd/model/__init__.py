"""Callers of the hot path restated on top of it: the ChronoSynth head (model/chrono_synth.py
upstream) running the fused head op, and the ArTEMIS model around it on stock PyTorch ops
(model/artemis.py, model/sep_sts_encoder.py, model/sep_sts_layer.py, model/helper_modules.py) --
all state-dict compatible with upstream."""
from .chrono_synth import ChronoSynth  # noqa: F401
from .encoder import SepSTSEncoder  # noqa: F401
from .artemis import ArTEMIS  # noqa: F401
