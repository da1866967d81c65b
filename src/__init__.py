"""Import shim: ``from src.models.cabinet import CABiNet`` (the reference's module path,
reference src/scripts/train.py:19, evaluate.py:18) resolves to the MI355X-native mirror.

``src`` is the reference's own top-level package name, so this package must not SHADOW it: with
``PYTHONPATH=<this repo>:<reference checkout>`` the reference's scripts import, next to the model,
``src.datasets.registry``, ``src.utils.{optimizer,ema,early_stopping,...}`` and ``src.scripts.evaluate``
(train.py:18-31).  ``pkgutil.extend_path`` appends every other ``src/`` directory on ``sys.path`` to this
package's search path: modules that exist here (``src.models.{cab,cabinet,constants,mobilenetv3}``,
``src.utils.loss``) win because this repo comes first, every other ``src.*`` module resolves to the reference.
``tests/test_reference_drives_it.py`` runs the reference's unmodified ``train_and_evaluate`` this way."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
