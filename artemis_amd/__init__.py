"""artemis_amd: the ArTEMIS frame-synthesis hot path (cupy_module/synth.py upstream) as
hand-written MI355X (gfx950) HIP kernels behind a C ABI, with the reference's
``FunctionSynth`` operator interface on top.  See DESIGN.md / INTEGRATION.md."""
from .synth import (FunctionSynth, FunctionSynthFrame, synth, synth_unpadded, synth_strip,  # noqa: F401
                    synth_options, options, offset_radius, suggest_tile_radius, adaptive_state,
                    FunctionChronoHead,
                    chrono_head)
from ._lib import version, SynthLibraryError  # noqa: F401

__all__ = ["FunctionSynth", "FunctionSynthFrame", "synth", "synth_unpadded", "synth_strip", "FunctionChronoHead", "chrono_head", "synth_options", "options", "offset_radius", "suggest_tile_radius", "adaptive_state", "version",
           "SynthLibraryError"]
