from .trainer import CrossDomainTrainer  # noqa: F401
