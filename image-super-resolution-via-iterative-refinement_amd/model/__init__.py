"""Drop-in `model` package: same entry point as the reference (model/__init__.py:5-9)."""
import logging

logger = logging.getLogger('base')


def create_model(opt):
    from .model import DDPM
    m = DDPM(opt)
    logger.info('Model [{:s}] is created.'.format(m.__class__.__name__))
    return m
