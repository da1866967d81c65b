"""Shim package for the reference's ``src.models`` (see ``src/__init__.py``): the four model modules below resolve to
``cabinet_amd.models``; anything else under the reference's ``src/models`` (``layers/``) is found through the extended
search path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
