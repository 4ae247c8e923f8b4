"""Loss-weight ramp, restating scheduler/customized_scheduler.py:42-65 (TEST INFRASTRUCTURE ONLY)."""
import math


def ramp_value(epoch: int, begin_epoch: int, max_epoch: int, max_value: float, ramp_mult: float) -> float:
    if epoch < begin_epoch:
        return 0.0
    if epoch >= max_epoch:
        return float(max_value)
    t = 1.0 - float(epoch - begin_epoch) / (max_epoch - begin_epoch)
    return float(max_value) * math.exp(ramp_mult * t * t)
