from .adamw import FusedAdamW, get_lr_sched, warmup_linear, warmup_cosine  # noqa: F401
