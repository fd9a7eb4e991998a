"""Bounded plan cache: ladder of padded lengths + LRU with a byte budget."""
import torch

from qflux_amd.plan_cache import PlanCache, arena_bytes, ladder


class _FakePlan:
    def __init__(self, nbytes):
        t = torch.zeros(nbytes, dtype=torch.uint8)
        self.A = {"x": t, "alias": t[: nbytes // 2], "nested": [{"y": torch.zeros(16, dtype=torch.float32)}, (t,)]}


def test_ladder_and_arena_bytes():
    assert [ladder(n, 128) for n in (1, 128, 129, 800, 2048, 3200)] == [128, 128, 256, 896, 2048, 3200]
    assert ladder(777, 0) == 777
    assert arena_bytes(_FakePlan(1000).A) == 1000 + 64            # views / repeated references of one storage count once
    # cfg #5's continuum (clusters at 800 / 2048 / 3200 tokens +-6 %) maps to a handful of ladder sizes
    sizes = {ladder(int(c * (1 + d / 100)), 128) for c in (800, 2048, 3200) for d in range(-6, 7)}
    assert len(sizes) <= 10


def test_lru_eviction_by_bytes_and_entries():
    built = []

    def mk(n):
        def b():
            built.append(n)
            return _FakePlan(n)
        return b
    c = PlanCache(budget_bytes=2500, max_entries=8)
    a = c.get_or_build("a", mk(1000))
    assert c.get_or_build("a", mk(1000)) is a and built == [1000]
    c.get_or_build("b", mk(1000))
    assert list(c) == ["a", "b"] and c.total_bytes() == 2 * 1064
    c.get_or_build("a", mk(1000))                  # touch: "b" is now the least recently used
    c.get_or_build("c", mk(1000))                  # 3 x 1064 > 2500 -> evict "b"
    assert list(c) == ["a", "c"] and c.evictions == 1 and c.total_bytes() == 2 * 1064
    c.get_or_build("huge", mk(10_000))             # larger than the budget on its own: everything else goes, the new plan stays
    assert list(c) == ["huge"] and c.total_bytes() == 10_064
    c2 = PlanCache(budget_bytes=1 << 40, max_entries=3)
    for i in range(20):
        c2.get_or_build(i, mk(10))
    assert list(c2) == [17, 18, 19] and c2.builds == 20 and c2.evictions == 17
    del c2[18]
    assert c2.total_bytes() == 2 * (10 + 64)
