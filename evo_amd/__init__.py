"""evo_amd -- MI355X-native (gfx950) E-step / M-step hot path of EVO (tvlearn/evo).

Public surface mirrors the reference's operator API for this path:
``evo_amd.models.{BSC, SSSC}`` and ``evo_amd.variational.{init_states, evolve_states, vary_Kn}``.
Importing the package needs neither a GPU nor the native library; the first compute call does.
"""
__version__ = "0.1.0"
