"""Cache-only loader on CPU: the reference's on-disk layout, collate semantics, rank-strided sharding."""
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen-image-finetune_amd"))


def _make_cache(tmp, n=10):
    from qflux_amd.data import write_cache_sample
    g = torch.Generator().manual_seed(0)
    ref = {}
    for i in range(n):
        S = 24 if i % 2 == 0 else 16      # two buckets -> ragged batches
        T = 5 + (i % 3)
        t = dict(image_latents=torch.randn(S, 64, generator=g), control_latents=torch.randn(S, 64, generator=g),
                 prompt_embeds=torch.randn(T, 32, generator=g), prompt_embeds_mask=torch.ones(T),
                 empty_prompt_embeds=torch.zeros(2, 32), empty_prompt_embeds_mask=torch.ones(2))
        h = f"{i:032x}"
        write_cache_sample(tmp, h, t, img_shapes=[(3, 64, 96), (3, 64, 96)], hashes={"prompt_embeds": h + "p", "empty_prompt_embeds": "e"})
        ref[h] = t
    return ref


def test_layout_roundtrip_and_collate(tmp_path):
    from qflux_amd.data import CachedEmbeddingDataset, collate_cached, pad_to_max_shape
    ref = _make_cache(tmp_path)
    # layout of EmbeddingCacheManager v2.0 (cache_manager.py:40-93)
    assert os.path.exists(tmp_path / "metadata" / f"{0:032x}.json")
    assert os.path.exists(tmp_path / "prompt_embeds" / (f"{0:032x}p.pt"))
    ds = CachedEmbeddingDataset(str(tmp_path))
    assert len(ds) == 10
    it = ds[3]
    want = ref[it["main_hash"]]
    assert it["image_latents"].dtype == torch.float16 and torch.equal(it["image_latents"], want["image_latents"].half())
    assert "empty_prompt_embeds" not in it and it["img_shapes"] == [(3, 64, 96), (3, 64, 96)] and it["cached"] is True
    b = collate_cached([ds[0], ds[1], ds[2]])
    assert b["image_latents"].shape == (3, 24, 64) and b["prompt_embeds"].shape == (3, 7, 32)
    assert torch.all(b["image_latents"][1, 16:] == 0) and torch.all(b["prompt_embeds_mask"][0, 5:] == 0)   # right padding with zeros
    assert len(b["img_shapes"]) == 3
    x = pad_to_max_shape([torch.ones(2, 3), torch.zeros(4, 1)])
    assert x.shape == (2, 4, 3) and x[0, :2].eq(1).all() and x[0, 2:].eq(0).all()
    # caption dropout swaps in the cached empty-prompt embeddings (dataset.py:548-554)
    dd = CachedEmbeddingDataset(str(tmp_path), caption_dropout_rate=1.0, prompt_empty_drop_keys=("empty_prompt_embeds", "empty_prompt_embeds_mask"))
    assert dd[0]["prompt_embeds"].shape == (2, 32)


def test_prefetch_loader_shards_disjointly(tmp_path):
    from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader
    _make_cache(tmp_path, n=12)
    ds = CachedEmbeddingDataset(str(tmp_path))
    seen = []
    for r in range(2):
        ld = PrefetchLoader(ds, batch_size=2, device="cpu", rank=r, world=2, seed=7, workers=3)
        ld.set_epoch(1)
        got = [h for b in ld for h in b["main_hash"]]
        assert len(got) == 6 and len(ld) == 3
        seen.append(set(got))
    assert not (seen[0] & seen[1]) and len(seen[0] | seen[1]) == 12
    # a different epoch reshuffles, the same epoch is reproducible
    ld = PrefetchLoader(ds, batch_size=2, device="cpu", seed=7)
    a = [h for b in ld for h in b["main_hash"]]
    b_ = [h for b in ld for h in b["main_hash"]]
    ld.set_epoch(3)
    c = [h for b in ld for h in b["main_hash"]]
    assert a == b_ and a != c and sorted(a) == sorted(c)


def test_every_rank_gets_the_same_number_of_batches(tmp_path):
    """10 samples over 4 ranks used to give 3,3,2,2 batches: the ranks with the extra batch then hang in the step's collectives.
    The reference's accelerate-prepared DataLoader yields the same count on every rank (base_trainer.py:378-393)."""
    from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader
    _make_cache(tmp_path, n=10)
    ds = CachedEmbeddingDataset(str(tmp_path))
    for world, bs, drop_last, want in ((4, 1, True, 2), (4, 1, False, 3), (3, 2, True, 1), (3, 2, False, 2), (2, 5, True, 1)):
        lens, seen = [], []
        for r in range(world):
            ld = PrefetchLoader(ds, batch_size=bs, device="cpu", rank=r, world=world, seed=3, workers=2, drop_last=drop_last)
            got = [h for b in ld for h in b["main_hash"]]
            lens.append(len(ld))
            assert len(got) == len(ld) * bs
            seen += got
        assert lens == [want] * world, (world, bs, drop_last, lens)
        if drop_last:
            assert len(set(seen)) == len(seen)               # disjoint shards
        else:
            assert len(set(seen)) == 10                      # wrap-around padding covers every sample


def test_prefetch_window_bounds_the_staged_batches(tmp_path):
    """The readers must not run ahead of the consumer by more than `prefetch` batches (pinned host memory is bounded)."""
    import time
    from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader
    _make_cache(tmp_path, n=24)
    ds = CachedEmbeddingDataset(str(tmp_path))
    for prefetch in (1, 3):
        ld = PrefetchLoader(ds, batch_size=1, device="cpu", seed=1, workers=6, prefetch=prefetch)
        n = 0
        for _ in ld:
            time.sleep(0.01)      # a slow consumer: fast readers would otherwise stage the whole epoch
            n += 1
        assert n == 24 and 1 <= ld.stats["staged_max"] <= prefetch, ld.stats


def test_loader_surfaces_reader_errors(tmp_path):
    from qflux_amd.data import CachedEmbeddingDataset, PrefetchLoader
    _make_cache(tmp_path, n=4)
    ds = CachedEmbeddingDataset(str(tmp_path))
    os.remove(glob.glob(str(tmp_path / "image_latents" / "*.pt"))[0])
    ld = PrefetchLoader(ds, batch_size=1, device="cpu", shuffle=False, workers=2)
    try:
        list(ld)
    except FileNotFoundError:
        return
    raise AssertionError("a missing cache file must raise in the consumer, not hang it")
