"""Synthetic weights/inputs live in vampnet_amd/synth.py (they are data generators, not oracle logic); re-exported
here for the tests."""
from vampnet_amd.synth import *  # noqa: F401,F403
from vampnet_amd.synth import COARSE_DIMS, C2F_DIMS, TINY_COARSE_DIMS, TINY_C2F_DIMS  # noqa: F401
