"""Enums and plugin lookup, mirroring recbole_cdr/utils/{enum_type,utils}.py (enum_type.py:18-45, utils.py:16-59)."""
import importlib
from enum import Enum


class ModelType(Enum):
    CROSSDOMAIN = 1


class InputType(Enum):          # recbole.utils.InputType (third-party), same values
    POINTWISE = 1
    PAIRWISE = 2
    LISTWISE = 3


class CrossDomainDataLoaderState(Enum):
    BOTH = 1
    SOURCE = 2
    TARGET = 3
    OVERLAP = 4


train_mode2state = {'BOTH': CrossDomainDataLoaderState.BOTH, 'SOURCE': CrossDomainDataLoaderState.SOURCE,
                    'TARGET': CrossDomainDataLoaderState.TARGET, 'OVERLAP': CrossDomainDataLoaderState.OVERLAP}


def get_model(model_name):
    """Model class by name: module = lower-cased file name under model/cross_domain_recommender (utils.py:16-40)."""
    pkg = __name__.rsplit('.', 1)[0]
    path = f'{pkg}.model.cross_domain_recommender.{model_name.lower()}'
    try:
        module = importlib.import_module(path)
    except ModuleNotFoundError:
        raise ValueError('`model_name` [{}] is not the name of an existing model.'.format(model_name))
    return getattr(module, model_name)


def get_trainer(model_type, model_name):
    """`<Model>Trainer` if the trainer package has one, else CrossDomainTrainer (utils.py:43-59)."""
    pkg = __name__.rsplit('.', 1)[0]
    trainer_mod = importlib.import_module(f'{pkg}.trainer')
    return getattr(trainer_mod, model_name + 'Trainer', trainer_mod.CrossDomainTrainer)


def total_loss(losses):
    """The scalar a trainer backpropagates: ``losses`` itself, or the sum of a tuple's parts (the reference's ``sum(losses)``,
    trainer.py:55-63) -- added left to right WITHOUT Python sum()'s leading ``0 +`` (a launch of its own inside a captured step; adding
    0.0 changes no bit)."""
    if not isinstance(losses, tuple):
        return losses
    out = losses[0]
    for part in losses[1:]:
        out = out + part
    return out
