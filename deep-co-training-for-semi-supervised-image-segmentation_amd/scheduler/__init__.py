"""Epoch-stepped loss-weight schedules: host scalars that feed the step's total loss (cotraining_totalloss.py:246).
Same constructor arguments (the YAML's keys), ``value`` / ``step`` / ``state_dict`` and the same closed forms as the
reference (generalframework/scheduler/customized_scheduler.py:42-116); the forms are written once, as functions of the
epoch, and the static methods the reference exposes delegate to them."""
import math

__all__ = ["RampScheduler", "ConstantScheduler", "RampDownScheduler"]


def _gaussian_ramp(progress: float, mult: float) -> float:
    """exp(mult * (1 - progress)^2): with mult < 0 it rises from exp(mult) at progress 0 to 1 at progress 1"""
    rest = 1.0 - progress
    return math.exp(mult * rest * rest)


class _EpochScheduler:
    """Counts epochs; subclasses turn the count into ``value``.  The whole attribute dictionary is the state."""

    def __init__(self):
        self.epoch = 0

    def step(self):
        self.epoch += 1

    def state_dict(self):
        return {k: v for k, v in vars(self).items() if k != 'optimizer'}

    def load_state_dict(self, state_dict):
        vars(self).update(state_dict)


class RampScheduler(_EpochScheduler):
    """0 until ``begin_epoch``, ``max_value`` from ``max_epoch`` on, a Gaussian ramp between the two (:55-65)."""

    def __init__(self, begin_epoch, max_epoch, max_value, ramp_mult):
        super().__init__()
        self.begin_epoch, self.max_epoch = int(begin_epoch), int(max_epoch)
        self.max_value, self.mult = float(max_value), float(ramp_mult)

    @property
    def value(self):
        return self.get_lr(self.epoch, self.begin_epoch, self.max_epoch, self.max_value, self.mult)

    @staticmethod
    def get_lr(epoch, begin_epoch, max_epochs, max_val, mult):
        span = max_epochs - begin_epoch
        done = epoch - begin_epoch
        if done < 0:
            return 0.0
        return max_val if done >= span else max_val * _gaussian_ramp(done / span, mult)


class ConstantScheduler(_EpochScheduler):
    """0 until ``begin_epoch``, ``max_value`` afterwards"""

    def __init__(self, begin_epoch, max_value=1.0):
        super().__init__()
        self.begin_epoch, self.max_value = int(begin_epoch), float(max_value)

    @property
    def value(self):
        return self.get_lr(self.epoch, self.begin_epoch, self.max_value)

    @staticmethod
    def get_lr(epoch, begin_epoch, max_value):
        return max_value if epoch >= begin_epoch else 0.0


class RampDownScheduler(_EpochScheduler):
    """``max_value`` at epoch 0, ``min_val`` from ``cutoff`` on, max - max * ramp + min in between (:96-116; the reference's order of
    operations, so that the values agree to the last bit for any ``max_value``)."""

    def __init__(self, max_epoch, max_value, ramp_mult, min_val, cutoff):
        super().__init__()
        self.max_epoch, self.cutoff = int(max_epoch), int(cutoff)
        self.max_value, self.mult, self.min_val = float(max_value), float(ramp_mult), float(min_val)

    @property
    def value(self):
        return self.ramp_down(self.epoch, self.max_epoch, self.max_value, self.mult, self.min_val, self.cutoff)

    @staticmethod
    def ramp_down(epoch, max_epochs, max_val, mult, min_val, cutoff):
        if not cutoff < max_epochs:
            raise AssertionError("cutoff must lie before max_epochs")
        if epoch <= 0:
            return max_val
        if epoch >= cutoff:
            return min_val
        return max_val - max_val * _gaussian_ramp(epoch / cutoff, mult) + min_val
