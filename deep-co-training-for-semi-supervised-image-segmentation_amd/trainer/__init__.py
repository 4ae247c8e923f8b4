from .trainer import Trainer  # noqa: F401
from .cotraining_totalloss import CoTrainer  # noqa: F401
