"""Framework-free handlers of the reference's REST search route.

The reference serves ``POST /api/v1/vectors/search`` with FastAPI (wdbx/api/server.py:141-152, :353-364): a
``SearchModel`` body (:109-113, :321-325) is passed to ``wdbx.vector_search_async`` and the hits come back as
``{"results": [{"vector_id", "similarity", "metadata"}]}``.  The HTTP shell (uvicorn, auth, CORS) is control plane and
out of scope; these two coroutines are what a route function calls, with the same request and response shapes and the
same validation outcome (a malformed body is a 422 there; here a ``ValueError`` the caller maps to it):

    @router.post("/vectors/search")
    async def search_vectors(body: dict):
        return await search_endpoint(wdbx, body)

``search_batch_endpoint`` is the batch form the reference lacks: many queries in one request, answered by
one batched pass per shard (``vector_search_batch``).  Concurrent single requests need no batch route: they are
coalesced at ``VectorStore.search_async`` (what the reference's server produces, api/server.py:143).
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional

_FIELDS = ("query_vector", "limit", "threshold", "filter_metadata")


def _parse_common(payload: Dict[str, Any]):
    if not isinstance(payload, dict):
        raise ValueError("request body must be an object")
    limit = payload.get("limit", 10)
    threshold = payload.get("threshold", 0.0)
    flt = payload.get("filter_metadata")
    limit = 10 if limit is None else limit        # Optional[int] = 10 (server.py:111): null means the default
    threshold = 0.0 if threshold is None else threshold
    if isinstance(limit, bool) or not isinstance(limit, int):
        raise ValueError("limit must be an integer")
    if isinstance(threshold, bool) or not isinstance(threshold, (int, float)):
        raise ValueError("threshold must be a number")
    if flt is not None and not isinstance(flt, dict):
        raise ValueError("filter_metadata must be an object")
    return limit, float(threshold), flt


def _vector(v: Any, what: str) -> List[float]:
    if not isinstance(v, (list, tuple)) or not all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in v):
        raise ValueError(f"{what} must be a list of numbers")
    return [float(x) for x in v]


def _render(results) -> Dict[str, Any]:
    return {"results": [{"vector_id": vid, "similarity": sim, "metadata": meta} for vid, sim, meta in results]}


async def search_endpoint(wdbx, payload: Dict[str, Any]) -> Dict[str, Any]:
    """``POST /api/v1/vectors/search`` (server.py:141-152): body ``{"query_vector": [...], "limit": 10, "threshold": 0.0,
    "filter_metadata": null}`` -> ``{"results": [{"vector_id", "similarity", "metadata"}, ...]}``."""
    limit, threshold, flt = _parse_common(payload)
    if "query_vector" not in payload:
        raise ValueError("query_vector is required")
    query = _vector(payload["query_vector"], "query_vector")
    return _render(await wdbx.vector_search_async(query, limit, threshold, flt))


async def search_batch_endpoint(wdbx, payload: Dict[str, Any]) -> Dict[str, Any]:
    """Batch form (extension): body ``{"query_vectors": [[...], ...], "limit", "threshold", "filter_metadata"}`` ->
    ``{"results": [<one search_endpoint result list per query>]}``.  One batched matrix-core pass per shard."""
    import asyncio

    limit, threshold, flt = _parse_common(payload)
    if "query_vectors" not in payload or not isinstance(payload["query_vectors"], (list, tuple)):
        raise ValueError("query_vectors is required and must be a list of vectors")
    queries = [_vector(v, f"query_vectors[{i}]") for i, v in enumerate(payload["query_vectors"])]
    if not queries:
        return {"results": []}
    loop = asyncio.get_running_loop()
    per_query = await loop.run_in_executor(None, lambda: wdbx.vector_search_batch(queries, limit, threshold, flt))
    return {"results": [_render(r)["results"] for r in per_query]}
