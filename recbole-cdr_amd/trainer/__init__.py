from .trainer import CrossDomainTrainer, DCDCSRTrainer  # noqa: F401
