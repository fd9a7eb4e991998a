"""Bounded cache of launch plans.

A plan owns a persistent HBM arena (every saved activation of the step: ~4.5 MB per token at the 60-block Qwen size, ~11 GB at
S = 2432), so the cache must not grow with the number of distinct shapes a training run meets.  Multi-resolution batches
(BASELINE.json config #5) arrive with a continuum of padded lengths (token counts cluster around the
bucket areas, +-6 %), so

  * multi-resolution plans are built for a LADDER of image-token counts (S_i rounded up to a multiple of `ladder_step()`,
    default 128 tokens = half a GEMM row tile): the extra rows are ordinary padded rows of the masked launch program
    (row-masked to exact zeros, key-masked out of the attention), results are unchanged;
  * the cache itself is LRU with a byte budget and an entry cap: the least recently used plans are dropped (their arena goes
    back to the caching allocator and is reused by the next plan) until the survivors fit.

Environment: QFX_PLAN_LADDER (tokens, 0 = exact shapes), QFX_PLAN_CACHE_GB (default: 40 % of the device memory, at most 96),
QFX_PLAN_CACHE_MAX (entries, default 16).
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch


def ladder_step() -> int:
    return int(os.environ.get("QFX_PLAN_LADDER", "128"))


def ladder(n: int, step: int | None = None) -> int:
    """Smallest ladder size >= n."""
    step = ladder_step() if step is None else step
    if step <= 0:
        return int(n)
    return (int(n) + step - 1) // step * step


def arena_bytes(obj, _seen=None) -> int:
    """Bytes of every distinct tensor storage reachable from a plan's arena (nested dicts / lists / tuples of tensors)."""
    seen = set() if _seen is None else _seen
    if isinstance(obj, torch.Tensor):
        st = obj.untyped_storage()
        key = st.data_ptr()
        if key in seen:
            return 0
        seen.add(key)
        return st.nbytes()
    if isinstance(obj, dict):
        return sum(arena_bytes(v, seen) for v in obj.values())
    if isinstance(obj, (list, tuple)):
        return sum(arena_bytes(v, seen) for v in obj)
    return 0


class PlanCache(OrderedDict):
    """key -> plan, least recently used first.  `get_or_build` is the only way entries come in."""

    def __init__(self, budget_bytes: int | None = None, max_entries: int | None = None):
        super().__init__()
        self._budget = budget_bytes
        self._max = max_entries
        self.sizes = {}
        self.builds = 0
        self.evictions = 0

    def budget_bytes(self) -> int:
        if self._budget is not None:
            return self._budget
        env = os.environ.get("QFX_PLAN_CACHE_GB")
        if env is not None:
            return int(float(env) * (1 << 30))
        total = 0
        if torch.cuda.is_available():
            total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
        return min(96 << 30, int(0.4 * total)) if total else (96 << 30)

    def max_entries(self) -> int:
        return self._max if self._max is not None else int(os.environ.get("QFX_PLAN_CACHE_MAX", "16"))

    def total_bytes(self) -> int:
        return sum(self.sizes.values())

    def get_or_build(self, key, builder, sizer=None):
        if key in self:
            self.move_to_end(key)
            return self[key]
        # make room BEFORE building: the newcomer is estimated at the size of the largest resident plan of the same key class
        # (first element of the key: shared vs "multires"), so the peak is the budget, not the budget plus one plan
        cls = key[0] if isinstance(key, tuple) and key else key
        est = max([sz for k, sz in self.sizes.items() if (k[0] if isinstance(k, tuple) and k else k) == cls] or [0])
        self._evict(keep=None, incoming=est)
        try:
            plan = builder()
        except torch.OutOfMemoryError:
            # one retry with everything else gone (the estimate was too small, or weights / optimizer state ate the headroom)
            while len(self):
                del self[next(iter(self))]
                self.evictions += 1
            import gc
            gc.collect()
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
            plan = builder()
        self.builds += 1
        self[key] = plan
        self.sizes[key] = int((sizer or (lambda p: arena_bytes(getattr(p, "A", None))))(plan))
        self._evict(keep=key)
        return plan

    def _evict(self, keep, incoming: int = 0):
        """Drop least-recently-used plans until the survivors (plus `incoming` bytes and one entry, when a build is about to
        happen) fit the budget and the entry cap; `keep` is never dropped."""
        budget, cap = self.budget_bytes(), self.max_entries()
        n0 = self.evictions
        extra = 1 if (incoming or keep is None) else 0
        floor = 1 if keep is not None else 0
        while len(self) > floor and (len(self) + extra > cap or self.total_bytes() + incoming > budget):
            old = next(k for k in self if k != keep)
            del self[old]
            self.evictions += 1
        if self.evictions != n0:
            # a plan and its launch programs reference each other (bound methods / event closures in the call lists): only the
            # cycle collector returns the arena to the allocator.  Evictions are rare; collect right away.
            import gc
            gc.collect()

    def __delitem__(self, key):
        super().__delitem__(key)
        self.sizes.pop(key, None)

    def clear(self):
        super().clear()
        self.sizes.clear()
