"""Loss registry (reference: generalframework/loss/__init__.py:6-16)."""
from .loss import CrossEntropyLoss2d, JSD_2D, KL_Divergence_2D, Entropy_2D, softmax_channels  # noqa: F401
from . import loss as _loss_mod

__all__ = ['get_loss_fn']

LOSS = {'cross_entropy': CrossEntropyLoss2d,
        'jsd': JSD_2D}


def get_loss_fn(name: str, **kwargs):
    """'mse_2d' and 'partial_ce' of the reference registry are not on the co-training path
     and raise the same ValueError an unknown name raises there."""
    try:
        return LOSS.get(name)(**kwargs)
    except Exception as e:
        raise ValueError('name error when inputting the loss name, with %s' % str(e))


def set_debug_asserts(on: bool):
    _loss_mod.DEBUG_ASSERTS = bool(on)
