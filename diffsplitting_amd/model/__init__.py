"""Mirror of the reference ``model`` package (model/__init__.py:5-9)."""
import logging

logger = logging.getLogger("base")


def create_model(opt):
    from .model import DDPM as M
    m = M(opt)
    logger.info("Model [{:s}] is created.".format(m.__class__.__name__))
    return m
