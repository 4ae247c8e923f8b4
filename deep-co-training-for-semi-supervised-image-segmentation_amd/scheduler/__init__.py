"""Epoch-stepped loss-weight schedulers (host scalars feeding cotraining_totalloss.py:246).
Reference: generalframework/scheduler/customized_scheduler.py:42-116."""
import math

__all__ = ["RampScheduler", "ConstantScheduler", "RampDownScheduler"]


class _EpochScheduler(object):
    def __init__(self):
        self.epoch = 0

    def step(self):
        self.epoch += 1

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)


class RampScheduler(_EpochScheduler):
    """0 before begin_epoch, max_value from max_epoch on, max*exp(mult*(1-t)^2) in between (:55-65)."""

    def __init__(self, begin_epoch, max_epoch, max_value, ramp_mult):
        super().__init__()
        self.begin_epoch = int(begin_epoch)
        self.max_epoch = int(max_epoch)
        self.max_value = float(max_value)
        self.mult = float(ramp_mult)

    @property
    def value(self):
        return self.get_lr(self.epoch, self.begin_epoch, self.max_epoch, self.max_value, self.mult)

    @staticmethod
    def get_lr(epoch, begin_epoch, max_epochs, max_val, mult):
        if epoch < begin_epoch:
            return 0.
        if epoch >= max_epochs:
            return max_val
        t = 1. - float(epoch - begin_epoch) / (max_epochs - begin_epoch)
        return max_val * math.exp(mult * t ** 2)


class ConstantScheduler(_EpochScheduler):
    def __init__(self, begin_epoch, max_value=1.0):
        super().__init__()
        self.begin_epoch = int(begin_epoch)
        self.max_value = float(max_value)

    @property
    def value(self):
        return self.get_lr(self.epoch, self.begin_epoch, self.max_value)

    @staticmethod
    def get_lr(epoch, begin_epoch, max_value):
        return 0.0 if epoch < begin_epoch else max_value


class RampDownScheduler(_EpochScheduler):
    def __init__(self, max_epoch, max_value, ramp_mult, min_val, cutoff):
        super().__init__()
        self.max_epoch = int(max_epoch)
        self.max_value = float(max_value)
        self.mult = float(ramp_mult)
        self.min_val = float(min_val)
        self.cutoff = int(cutoff)

    @property
    def value(self):
        return self.ramp_down(self.epoch, self.max_epoch, self.max_value, self.mult, self.min_val, self.cutoff)

    @staticmethod
    def ramp_down(epoch, max_epochs, max_val, mult, min_val, cutoff):
        assert cutoff < max_epochs
        if epoch == 0:
            return max_val
        if epoch >= cutoff:
            return min_val
        return max_val - max_val * math.exp(mult * (1. - float(epoch) / cutoff) ** 2) + min_val
