"""wdbx_amd -- MI355X-native drop-in for the WDBX ``vector_search`` hot path.

Same Python surface as the reference for this path (``WDBX``, ``VectorStore``,
the ``VectorIndex`` backend seam, ``WDBXConfig``); the distance + top-k scan and
the shard merge run in hand-written HIP kernels behind a C ABI
(``include/wdbx_hip.h``) called through ctypes.  There is no CPU fallback.
"""

from .config import WDBXConfig
from .indexing import HipFlatIndex, VectorIndex
from .vector_store import VectorStore
from .wdbx import WDBX
from ._native import HipBackendError

__version__ = "0.1.0"

__all__ = ["WDBX", "VectorStore", "VectorIndex", "HipFlatIndex", "WDBXConfig", "HipBackendError", "__version__"]
