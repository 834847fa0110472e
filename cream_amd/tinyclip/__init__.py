"""TinyCLIP affinity-mimicking distillation loss on RCCL (first piece of BASELINE config 5)."""
from .soft_loss import ClipSoftLoss, gather_features_with_grad  # noqa: F401
