"""Drop-in for the reference's ``cupy_module/synth.py``: put ``artemis_amd/compat`` ahead of the
reference checkout on ``sys.path`` (or copy this two-line file over the original) and
``model/chrono_synth.py:66-67`` -- ``import cupy_module.synth as synth;
self.moduleSynth = synth.FunctionSynth.apply`` -- binds the MI355X kernels with no CuPy import.
"""
from artemis_amd.synth import FunctionSynth  # noqa: F401
