"""Mirror of the reference's ``tasks`` package for the hot path's caller: ``Trainer`` (tasks/trainer.py)."""
from .trainer import Trainer, NNDataLoader  # noqa: F401
