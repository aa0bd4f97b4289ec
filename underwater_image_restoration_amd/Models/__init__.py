"""Model factory with the reference's surface (src/Models/__init__.py:10-30): ``init_model``,
``get_names``.  ``SpectralTransformer`` (the MI355X hot path), ``NewBigFRFNModel`` and ``AST`` are built
here; ``NewModel`` / ``NewBigModel`` crash as shipped in the reference and raise with a clear message."""
from .SpectralTransformer import SpectralTransformer
from ..model.model import MyBigFRFNModel as NewBigFRFNModel
from .AST import AST

__model_factory = {
    "SpectralTransformer": SpectralTransformer,
    "NewBigFRFNModel": NewBigFRFNModel,
    "AST": AST,
}
_REFERENCE_ONLY = ("NewModel", "NewBigModel")
_ORDER = ("SpectralTransformer", "NewModel", "NewBigModel", "NewBigFRFNModel", "AST")       # src/Models/__init__.py:10-16


def get_names():
    return list(_ORDER)


def init_model(name, *args, **kwargs):
    print("modelCreated")
    if name in _REFERENCE_ONLY:
        raise NotImplementedError(f"{name}: not part of the MI355X hot-path build (see DESIGN.md, out of scope)")
    if name not in __model_factory:
        raise KeyError(f"Unknown model: {name}")
    kwargs.pop("use_dwt", None)   # the reference drops it for every arch but NewModel (__init__.py:25-29)
    return __model_factory[name](*args, **kwargs)
