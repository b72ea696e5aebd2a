"""MI355X-native bipartite GCNN cut scorer: the hot path of stefanvanberkum/gcnn-cut-selector's `model.py`
(GCNN forward/backward) as hand-written HIP for gfx950 behind the reference's Python call surface.

Import as `gcnn_cut_selector_amd` (see the shim module at the repository root)."""

__version__ = "0.1.0"
