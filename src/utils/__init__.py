"""Shim package for the reference's ``src.utils`` (see ``src/__init__.py``): ``src.utils.loss`` resolves to
``cabinet_amd.loss``; the reference's other utilities (optimizer, ema, early_stopping, logger, class_weights,
exceptions, profiler) are found through the extended search path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
