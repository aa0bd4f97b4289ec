"""MI355X-native SpectralTransformer training path (HIP kernels behind the reference's surface)."""
from . import _lib  # noqa: F401  (raises if libuir_hip.so is missing: no fallback)
from .Models import init_model, get_names  # noqa: F401
