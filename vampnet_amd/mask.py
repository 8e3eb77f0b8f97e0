"""`vampnet_amd.mask` = the reference's module name for the mask functions (`from vampnet import mask as pmask`, app.py:17,
scripts/exp/train.py:19); everything lives in vampnet_amd/masks.py."""
from .masks import *                     # noqa: F401,F403
from .masks import _gamma, _invgamma     # noqa: F401
