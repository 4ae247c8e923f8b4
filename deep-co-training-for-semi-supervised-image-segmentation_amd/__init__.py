"""dct_amd -- MI355X-native co-training step behind the reference's Python API.

Mirrors the reference package layout for the hot path only
(/root/reference/generalframework): ``ModelMode`` (generalframework/__init__.py:12-33),
``trainer.CoTrainer``, ``models.Segmentator``, ``loss.*``, ``utils.AEGenerator.FSGMGenerator``,
``arch.get_arch``, ``scheduler.*``, ``metrics.DiceMeter``.  All arithmetic on the path runs in
hand-written HIP kernels (csrc/, C ABI in include/dct.h); there is no CPU fallback.
"""
from enum import Enum

name = "dct_amd"
__version__ = "0.1.0"


class ModelMode(Enum):
    """generalframework/__init__.py:12-33."""
    TRAIN = 'TRAIN'
    EVAL = 'EVAL'
    PRED = 'PRED'

    @staticmethod
    def from_str(mode_str):
        table = {'train': ModelMode.TRAIN, 'eval': ModelMode.EVAL, 'predict': ModelMode.PRED}
        if mode_str not in table:
            raise ValueError('Invalid argument mode_str {}'.format(mode_str))
        return table[mode_str]
