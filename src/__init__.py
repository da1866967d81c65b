"""Import shim: ``from src.models.cabinet import CABiNet`` (the reference's module path,
reference src/scripts/train.py:19, evaluate.py:18) resolves to the MI355X-native mirror."""
