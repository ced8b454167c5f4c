"""sorrel_amd -- MI355X-native batched gridworld step/observation engine behind
Sorrel's Environment / Gridworld / Entity / Agent / ObservationSpec API.

The hot path (``Environment.take_turn`` and everything under it) runs in
hand-written HIP (``sorrel_amd/csrc/``) through the C ABI of
``include/sgw.h``; this package is the host-side mirror of the reference's
plugin interface.
"""
__version__ = "0.1.0"
