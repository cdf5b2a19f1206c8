"""Drop-in `model` package.  `create_model(opt)` is the entry point sr.py / infer.py / sample.py call
(reference model/__init__.py:5-9): it builds the DDPM wrapper around the MI355X engine and logs its class name."""
import logging


def create_model(opt):
    from . import model as _model
    wrapper = _model.DDPM(opt)
    logging.getLogger('base').info('Model [%s] is created.' % type(wrapper).__name__)
    return wrapper
