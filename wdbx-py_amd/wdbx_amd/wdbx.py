"""The ``WDBX`` facade for the hot path (reference: wdbx/core/wdbx.py:21-502):
constructor signature, ``initialize/shutdown``, ``vector_store[_async]``,
``vector_search[_async]``, row management and ``get_stats`` keep the reference's
names, arguments, return shapes and error text.  Plugins and the TCP
``ShardManager`` are out of scope; ``enable_plugins`` /
``enable_distributed`` are accepted and recorded only."""

from __future__ import annotations

import logging
import uuid
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

from .config import WDBXConfig
from .vector_store import VectorStore

logger = logging.getLogger(__name__)

Result = Tuple[str, float, Dict[str, Any]]


class WDBX:
    def __init__(
        self,
        vector_dimension: int = 384,
        num_shards: int = 1,
        data_dir: str = "./wdbx_data",
        config: Optional[Dict[str, Any]] = None,
        enable_plugins: bool = True,
        enable_distributed: bool = False,
        enable_gpu: bool = False,
        log_level: str = "INFO",
    ):
        """Same signature and defaults as the reference (wdbx.py:36-46).  ``enable_gpu`` is plumbed to the store and
        its indices exactly as there (:124); this backend always runs on the GPU -- see ``HipFlatIndex.__init__`` for
        what the flag means here (recorded, reported in ``get_stats`` as given, one warning when it is False)."""
        level = getattr(logging, log_level.upper(), None)
        if not isinstance(level, int):
            raise ValueError(f"Invalid log level: {log_level}")
        if not logging.getLogger().handlers:
            logging.basicConfig(level=level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s",
                                datefmt="%Y-%m-%d %H:%M:%S")
        self.vector_dim = vector_dimension
        self.num_shards = num_shards
        self.data_dir = Path(data_dir)
        self.config = WDBXConfig(config or {})
        self.enable_plugins = enable_plugins
        self.enable_distributed = enable_distributed
        self.enable_gpu = enable_gpu
        self.data_dir.mkdir(parents=True, exist_ok=True)

        # ``vector_store`` is the store object AND the store call (see VectorStore.__call__)
        self.vector_store = VectorStore(
            vector_dim=self.vector_dim,
            data_dir=self.data_dir,
            num_shards=self.num_shards,
            use_gpu=self.enable_gpu,
            index_type=self.config.get("INDEX_TYPE", "hip"),
            config=self.config,
        )
        self.plugins: Dict[str, Any] = {}
        self.shard_manager = None
        logger.info("WDBX initialized successfully with vector_dim=%d, num_shards=%d", self.vector_dim,
                    self.num_shards)

    @property
    def version(self) -> str:
        from . import __version__

        return __version__

    async def initialize(self):
        await self.vector_store.initialize()

    async def shutdown(self):
        await self.vector_store.shutdown()

    # ---- plugin registry (callers of the path; kept so embedding providers can register) ----
    def get_plugin(self, plugin_name: str):
        if not self.enable_plugins:
            return None
        return self.plugins.get(plugin_name)

    def register_plugin(self, plugin) -> bool:
        if not self.enable_plugins or plugin.name in self.plugins:
            return False
        self.plugins[plugin.name] = plugin
        return True

    # ---- store ----
    def _check_dim(self, vector) -> None:
        if len(vector) != self.vector_dim:
            raise ValueError(f"Vector dimension mismatch: expected {self.vector_dim}, got {len(vector)}")

    async def vector_store_async(self, vector: List[float], metadata: Optional[Dict[str, Any]] = None,
                                 id: Optional[str] = None) -> str:
        self._check_dim(vector)
        vector_id = id or str(uuid.uuid4())
        await self.vector_store.store_async(vector_id, vector, metadata)
        return vector_id

    # ---- search ----
    def vector_search(self, query_vector: List[float], limit: int = 10, threshold: float = 0.0,
                      filter_metadata: Optional[Dict[str, Any]] = None,
                      prefilter: Optional[bool] = None) -> List[Result]:
        self._check_dim(query_vector)
        return self.vector_store.search(query_vector, limit=limit, threshold=threshold,
                                        filter_metadata=filter_metadata, prefilter=prefilter)

    async def vector_search_async(self, query_vector: List[float], limit: int = 10, threshold: float = 0.0,
                                  filter_metadata: Optional[Dict[str, Any]] = None) -> List[Result]:
        self._check_dim(query_vector)
        return await self.vector_store.search_async(query_vector, limit=limit, threshold=threshold,
                                                    filter_metadata=filter_metadata)

    def vector_search_batch(self, query_vectors, limit: int = 10, threshold: float = 0.0,
                            filter_metadata: Optional[Dict[str, Any]] = None) -> List[List[Result]]:
        """Extension: several queries at once (the reference is single-query)."""
        for q in query_vectors:
            self._check_dim(q)
        return self.vector_store.search_batch(query_vectors, limit=limit, threshold=threshold,
                                              filter_metadata=filter_metadata)

    # ---- row management ----
    def delete_vector(self, vector_id: str) -> bool:
        return self.vector_store.delete(vector_id)

    async def delete_vector_async(self, vector_id: str) -> bool:
        return await self.vector_store.delete_async(vector_id)

    def update_metadata(self, vector_id: str, metadata: Dict[str, Any]) -> bool:
        return self.vector_store.update_metadata(vector_id, metadata)

    async def update_metadata_async(self, vector_id: str, metadata: Dict[str, Any]) -> bool:
        return await self.vector_store.update_metadata_async(vector_id, metadata)

    def get_vector(self, vector_id: str):
        return self.vector_store.get(vector_id)

    async def get_vector_async(self, vector_id: str):
        return await self.vector_store.get_async(vector_id)

    def count_vectors(self) -> int:
        return self.vector_store.count()

    def clear(self) -> int:
        return self.vector_store.clear()

    async def clear_async(self) -> int:
        return await self.vector_store.clear_async()

    def get_stats(self) -> Dict[str, Any]:
        stats = {
            "version": self.version,
            "vector_dimension": self.vector_dim,
            "num_shards": self.num_shards,
            "total_vectors": self.count_vectors(),
            "plugins_enabled": self.enable_plugins,
            "plugins_loaded": len(self.plugins) if self.enable_plugins else 0,
            "distributed_enabled": self.enable_distributed,
            "gpu_enabled": self.enable_gpu,
        }
        stats.update(self.vector_store.get_stats())
        return stats
