"""Callers of the hot path restated on top of it: the ChronoSynth head (model/chrono_synth.py
upstream), state-dict compatible, running the fused head op."""
from .chrono_synth import ChronoSynth  # noqa: F401
