"""Configuration for the hot path: same precedence and accessors as the reference's
``WDBXConfig`` (wdbx/core/config.py:14-314): defaults < JSON file < ``WDBX_*``
environment < runtime dict; ``get/get_typed/set/has`` and mapping access.
Keys the HIP backend adds, in the reference's un-prefixed style:
``INDEX_TYPE`` ("hip"), ``HIP_METRIC`` ("cosine" | "l2"), ``HIP_DEVICES`` (list of
device ids the shards are dealt over), ``HIP_CAPACITY_ROWS`` (initial rows per
shard), ``HIP_SWALLOW_ERRORS`` (default True = the reference's convention, indexing.py:1028-1030: a backend
error in add / search is logged and becomes ``False`` / ``[]``; False raises it), ``HIP_GROUP_SEARCH`` (several shards:
fan-out + exchange + merge in ONE library call, a host thread per shard; "auto" = for shards that share GPUs (exchange by device
copies), True = also one shard per GPU (RCCL all-gather), "always" = also a single shard, False = off), ``HIP_AUTOSAVE_ROWS``
(index files follow ingest every N adds, reference: 1000), ``HIP_PERSIST_INDEX``, ``HIP_COMPACT_MIN_FRACTION`` (``optimize()`` compacts removed rows away once they are this share of a shard; 0 = any), ``HIP_BF16_SHADOW`` / ``HIP_U8_SHADOW`` (keep a bf16 / u8 copy of the
rows for the batched / single-query selection passes: +50 % / +25 % device memory,
several times the query rate; results are the exact fp32 ranking either way), ``FILTER_PUSHDOWN`` (metadata filter before the scan),
``ASYNC_COALESCE`` (concurrent ``search_async`` callers share one batched pass)."""

from __future__ import annotations

import json
import logging
import os
from pathlib import Path
from typing import Any, Dict, Optional

logger = logging.getLogger(__name__)

_TRUE = ("true", "yes", "1", "on")
_FALSE = ("false", "no", "0", "off")


def _parse_env_value(text: str) -> Any:
    """JSON first, then bool / int / float spellings, else the string itself
    (config.py:129-156)."""
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        pass
    low = text.lower()
    if low in _TRUE:
        return True
    if low in _FALSE:
        return False
    if text.isdigit():
        return int(text)
    if text.count(".") <= 1 and text.replace(".", "", 1).isdigit():
        return float(text)
    return text


class WDBXConfig:
    DEFAULT_CONFIG: Dict[str, Any] = {
        "VECTOR_STORE_SAVE_IMMEDIATELY": False,
        "VECTOR_STORE_THREADS": os.cpu_count() or 4,
        "VECTOR_STORE_CACHE_SIZE_MB": 128,
        "HNSW_M": 16,
        "HNSW_EF_CONSTRUCTION": 200,
        "HNSW_EF_SEARCH": 50,
        "FAISS_INDEX_TYPE": "Flat",
        "FAISS_NPROBE": 8,
        "DISTRIBUTED_HOST": "localhost",
        "DISTRIBUTED_PORT": 7777,
        "DISTRIBUTED_AUTH_ENABLED": False,
        "DISTRIBUTED_AUTH_KEY": "",
        "PLUGIN_DIRECTORY": "plugins",
        "PLUGIN_AUTO_DISCOVER": True,
        "PLUGIN_TIMEOUT": 30,
        # HIP backend
        "INDEX_TYPE": "hip",
        "HIP_METRIC": "cosine",
        "HIP_DEVICES": None,
        "HIP_CAPACITY_ROWS": 4096,
        "HIP_SWALLOW_ERRORS": True,
        "HIP_BF16_SHADOW": True,
        "HIP_U8_SHADOW": True,
        "HIP_GROUP_SEARCH": "auto",
        "HIP_AUTOSAVE_ROWS": 1000,
        "HIP_PERSIST_INDEX": True,
        "HIP_COMPACT_MIN_FRACTION": 0.0,
        "FILTER_PUSHDOWN": False,
        "ASYNC_COALESCE": True,
    }

    def __init__(self, config_dict: Optional[Dict[str, Any]] = None, config_path: Optional[str] = None):
        self.config_dict: Dict[str, Any] = dict(self.DEFAULT_CONFIG)
        self.config_sources: Dict[str, str] = {k: "default" for k in self.DEFAULT_CONFIG}
        if config_path:
            self._load_config_from_file(config_path)
        self._load_config_from_env()
        for key, value in (config_dict or {}).items():
            self.config_dict[key] = value
            self.config_sources[key] = "runtime"

    def _load_config_from_file(self, config_path: str) -> None:
        path = Path(config_path)
        if not path.exists():
            logger.warning("Configuration file not found: %s", config_path)
            return
        try:
            with open(path, "r") as f:
                loaded = json.load(f)
        except Exception as e:  # malformed file: keep going with what we have
            logger.error("Error loading configuration from file %s: %s", config_path, e)
            return
        for key, value in loaded.items():
            self.config_dict[key] = value
            self.config_sources[key] = f"file:{config_path}"

    def _load_config_from_env(self) -> None:
        for key, value in os.environ.items():
            if key.startswith("WDBX_"):
                self.config_dict[key] = _parse_env_value(value)
                self.config_sources[key] = "environment"

    # accessors -------------------------------------------------------------
    def get(self, key: str, default: Any = None) -> Any:
        return self.config_dict.get(key, default)

    def set(self, key: str, value: Any) -> None:
        self.config_dict[key] = value
        self.config_sources[key] = "runtime"

    def has(self, key: str) -> bool:
        return key in self.config_dict

    def get_source(self, key: str) -> Optional[str]:
        return self.config_sources.get(key)

    def get_all(self) -> Dict[str, Any]:
        return dict(self.config_dict)

    def get_typed(self, key: str, expected_type: type, default: Any = None) -> Any:
        value = self.get(key, default)
        if value is None or isinstance(value, expected_type):
            return value
        try:
            if expected_type is bool:
                return value.lower() in _TRUE if isinstance(value, str) else bool(value)
            if expected_type in (int, float, str):
                return expected_type(value)
            if expected_type is list:
                if isinstance(value, str):
                    return json.loads(value) if value.startswith("[") else value.split(",")
                return list(value)
            if expected_type is dict:
                return json.loads(value) if isinstance(value, str) else dict(value)
        except (ValueError, TypeError, json.JSONDecodeError):
            pass
        logger.warning("Could not convert config value %s=%r to %s, using default", key, value,
                       expected_type.__name__)
        return default

    def save_to_file(self, config_path: str) -> bool:
        try:
            path = Path(config_path)
            path.parent.mkdir(parents=True, exist_ok=True)
            with open(path, "w") as f:
                json.dump(self.config_dict, f, indent=2, sort_keys=True)
            return True
        except Exception as e:
            logger.error("Error saving configuration to file %s: %s", config_path, e)
            return False

    def reset(self) -> None:
        self.config_dict = dict(self.DEFAULT_CONFIG)
        self.config_sources = {k: "default" for k in self.DEFAULT_CONFIG}

    def __getitem__(self, key: str) -> Any:
        return self.get(key)

    def __setitem__(self, key: str, value: Any) -> None:
        self.set(key, value)

    def __contains__(self, key: str) -> bool:
        return self.has(key)

    def __len__(self) -> int:
        return len(self.config_dict)

    def __repr__(self) -> str:
        return f"WDBXConfig({self.config_dict})"
