"""Cache-only input pipeline for the cached-embedding training step.

On-disk format = the reference's EmbeddingCacheManager v2.0 (src/qflux/data/cache_manager.py:40-125):
    <cache_root>/metadata/<main_hash>.json     {"version": "2.0", "<key>": "<hash>", ..., "img_shapes": [[C,H,W], ...]}
    <cache_root>/<key>/<hash>.pt               torch.save(fp16 tensor)         (keys: image_latents, control_latents,
                                                prompt_embeds, prompt_embeds_mask, pooled_prompt_embeds, empty_* ...)
The reference's dataset still decodes and preprocesses the images on every step even when the cache exists
(dataset.py:540-556: preprocess at :542 runs before load_cache at :555) and hands over unpinned tensors
(pin_memory=False, :748).  At ~10 images/s per GPU that is the bottleneck, so this loader
  * reads ONLY the metadata + .pt files of a sample (no image decode),
  * collates like the reference (right-pad every tensor to the batch maximum, dataset.py:641-695, tools.py:399-425),
  * stages each batch in PINNED host buffers from background threads and uploads it on a side HIP stream, one batch ahead;
    the consumer's stream waits on an event, never on the host.
Sharding: rank-strided indices over a per-epoch seeded permutation (every rank sees a disjoint slice; no collective).
"""
from __future__ import annotations

import glob
import json
import os
import random
import threading
import time

import torch
import torch.nn.functional as F

_SKIP = ("version", "img_shapes")


def pad_to_max_shape(tensors, padding_value=0):
    """Right-pad same-rank tensors to their per-dimension maximum and stack (src/qflux/utils/tools.py:399-425)."""
    max_shape = [max(s) for s in zip(*[t.shape for t in tensors])]
    out = []
    for t in tensors:
        pad = []
        for i in range(len(max_shape) - 1, -1, -1):
            pad.extend([0, max_shape[i] - t.shape[i]])
        out.append(F.pad(t, pad, value=padding_value))
    return torch.stack(out, dim=0)


def write_cache_sample(cache_root, main_hash, tensors: dict, img_shapes=None, hashes: dict | None = None):
    """Writer with the reference's layout (save_cache_embedding, cache_manager.py:48-93); used by tests and tools."""
    meta = {"version": "2.0"}
    for key, t in tensors.items():
        h = (hashes or {}).get(key, main_hash)
        path = os.path.join(str(cache_root), key, f"{h}.pt")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(t.detach().cpu().to(torch.float16), path)
        meta[key] = h
    if img_shapes is not None:
        meta["img_shapes"] = [list(s) for s in img_shapes]
    mp = os.path.join(str(cache_root), "metadata", f"{main_hash}.json")
    os.makedirs(os.path.dirname(mp), exist_ok=True)
    with open(mp, "w") as f:
        json.dump(meta, f, indent=2)


class CachedEmbeddingDataset:
    """One item = the cached tensors of one sample (load_cache, cache_manager.py:95-125), nothing else touched."""

    def __init__(self, cache_root: str, caption_dropout_rate: float = 0.0, prompt_empty_drop_keys=(), seed: int = 1234):
        self.cache_root = str(cache_root)
        self.metas = sorted(glob.glob(os.path.join(self.cache_root, "metadata", "*.json")))
        if not self.metas:
            raise FileNotFoundError(f"no cache metadata under {self.cache_root}/metadata (EmbeddingCacheManager.exist is False)")
        self.caption_dropout_rate = caption_dropout_rate
        self.prompt_empty_drop_keys = tuple(prompt_empty_drop_keys)
        self._rng = random.Random(seed)

    def __len__(self):
        return len(self.metas)

    def _load(self, key, h):
        return torch.load(os.path.join(self.cache_root, key, f"{h}.pt"), map_location="cpu", weights_only=False)

    def __getitem__(self, idx: int) -> dict:
        with open(self.metas[idx]) as f:
            meta = json.load(f)
        data = {"cached": True, "main_hash": os.path.splitext(os.path.basename(self.metas[idx]))[0]}
        for key, h in meta.items():
            if key in _SKIP or key.startswith("empty_"):
                continue
            data[key] = self._load(key, h)
        if "img_shapes" in meta:
            data["img_shapes"] = [tuple(int(v) for v in s) for s in meta["img_shapes"]]
        if self.prompt_empty_drop_keys and self._rng.random() < self.caption_dropout_rate:   # caption dropout (dataset.py:548-554)
            for key in self.prompt_empty_drop_keys:
                data[key.replace("empty_", "")] = self._load(key, meta[key])
        return data


def convert_img_shapes_to_latent_space(img_shapes, vae_scale_factor: int = 8):
    """[(C,H,W), ...] per sample in pixel space -> [(1, H/16, W/16), ...] (qwen_image_edit_trainer.py:557-577)."""
    return [[(1, s[1] // vae_scale_factor // 2, s[2] // vae_scale_factor // 2) for s in per] for per in img_shapes]


def collate_cached(batch: list) -> dict:
    """collate_fn of the reference restricted to cached samples (dataset.py:641-695): tensors right-padded and stacked, lists kept."""
    out = {}
    for key in batch[0].keys():
        vals = [b[key] for b in batch]
        out[key] = pad_to_max_shape(vals) if isinstance(vals[0], torch.Tensor) else vals
    return out


def _to_device(obj, device):
    """Tensors of a (possibly nested) batch value to the device, asynchronously; everything else unchanged."""
    if isinstance(obj, torch.Tensor):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, list):
        return [_to_device(v, device) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_to_device(v, device) for v in obj)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    return obj


def _record_stream(obj, stream):
    """record_stream on every CUDA tensor of a (possibly nested) batch container."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            if k != "_host":
                _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class PrefetchLoader:
    """Iterates device-resident batches.  `workers` threads read + collate into pinned memory; the upload of batch n+1 runs on a
    side stream while the training step consumes batch n."""

    def __init__(self, dataset, batch_size: int, device, rank: int = 0, world: int = 1, shuffle: bool = True, seed: int = 1234,
                 workers: int = 2, prefetch: int = 3, drop_last: bool = True, keys_to_device=None):
        self.ds, self.bs, self.device = dataset, batch_size, torch.device(device)
        self.rank, self.world, self.shuffle, self.seed = rank, world, shuffle, seed
        self.workers, self.prefetch, self.drop_last = workers, prefetch, drop_last
        self.keys_to_device = keys_to_device
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def indices(self):
        """Batches of this rank.  Every rank gets the SAME number of batches (the permutation is cut to a multiple of
        world * batch_size before striding -- or wrap-padded when drop_last is False, as accelerate's prepared DataLoader does,
        base_trainer.py:378-393): a rank with one batch more would sit alone in the step's collectives at the end of the epoch."""
        idx = list(range(len(self.ds)))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(idx)
        chunk = self.world * self.bs
        if self.drop_last or not idx:
            idx = idx[:len(idx) // chunk * chunk]
        elif len(idx) % chunk:
            idx = idx + (idx * chunk)[:chunk - len(idx) % chunk]   # wrap-around padding to a whole number of global batches
        # whole batches dealt round-robin: global batch j goes to rank j % world -- the composition accelerate's BatchSamplerShard
        # (split_batches=False, the reference's setting) gives each process for the same permutation
        batches = [idx[i:i + self.bs] for i in range(0, len(idx), self.bs)]
        return batches[self.rank::self.world]

    def __len__(self):
        return len(self.indices())

    def _host_batch(self, ids):
        b = collate_cached([self.ds[i] for i in ids])
        if self.device.type == "cuda":
            for k, v in b.items():
                if isinstance(v, torch.Tensor):
                    b[k] = v.pin_memory()
        return b

    def __iter__(self):
        batches = self.indices()
        depth = max(1, self.prefetch)
        slots = {}
        # one lock, two wait queues: a consumed batch frees ONE slot and wakes ONE reader; a staged batch wakes the consumer.  (With
        # a single condition + notify_all every take() woke all readers: four of them thrashing the GIL under the launch thread
        # cost the step 3 % -- tools/hostfed_probe.py.)
        lock = threading.Lock()
        space, ready = threading.Condition(lock), threading.Condition(lock)
        cv = ready
        state = {"next": 0, "consumed": 0, "stop": False, "staged_max": 0, "error": None, "wait_s": 0.0, "upload_s": 0.0}
        # staged_max = the largest number of host batches ever staged at once (tests read it); wait_s = time the consumer spent
        # blocked on a batch the readers had not finished; upload_s = host time of issuing the uploads
        self.stats = state

        def work():
            # back-pressure: batch i is read only once i < consumed + depth, so at most `depth` collated (pinned) batches exist
            # on the host at any time however fast the disk readers are
            while True:
                with lock:
                    while not state["stop"] and state["next"] < len(batches) and state["next"] >= state["consumed"] + depth:
                        space.wait()
                    if state["stop"] or state["next"] >= len(batches):
                        return
                    i = state["next"]
                    state["next"] += 1
                try:
                    hb = self._host_batch(batches[i])
                except BaseException as e:   # noqa: BLE001  (surfaced in the consumer thread)
                    with lock:
                        state["error"] = e
                        state["stop"] = True
                        space.notify_all()
                        ready.notify_all()
                    return
                with lock:
                    slots[i] = hb
                    state["staged_max"] = max(state["staged_max"], len(slots))
                    ready.notify_all()

        threads = [threading.Thread(target=work, daemon=True) for _ in range(min(self.workers, max(1, len(batches))))]
        for t in threads:
            t.start()
        use_cuda = self.device.type == "cuda"
        side = torch.cuda.Stream(device=self.device) if use_cuda else None

        def take(i):
            with cv:
                if i not in slots and state["error"] is None:
                    t_w = time.perf_counter()
                    while i not in slots and state["error"] is None:
                        cv.wait()
                    state["wait_s"] += time.perf_counter() - t_w
                if state["error"] is not None:
                    raise state["error"]
                hb = slots.pop(i)
                state["consumed"] = i + 1
                space.notify()
            return hb

        def upload(i):
            hb = take(i)
            if not use_cuda:
                return hb, None
            t_u = time.perf_counter()
            ev = torch.cuda.Event()
            with torch.cuda.stream(side):
                db = {k: (_to_device(v, self.device) if (self.keys_to_device is None or k in self.keys_to_device) else v)
                      for k, v in hb.items()}
                ev.record(side)
            db["_host"] = hb     # keep the pinned buffers alive until the copy has been consumed
            state["upload_s"] += time.perf_counter() - t_u
            return db, ev

        try:
            pending = upload(0) if batches else None
            for i in range(len(batches)):
                cur, ev = pending
                pending = upload(i + 1) if i + 1 < len(batches) else None    # next batch uploads while this one trains
                if ev is not None:
                    consumer = torch.cuda.current_stream(self.device)
                    consumer.wait_event(ev)
                    # the tensors were allocated on the side stream's pool: tell the caching allocator that the consumer stream
                    # uses them too, or their blocks could be handed to upload(i+2) while this step's kernels are still queued
                    _record_stream(cur, consumer)
                cur.pop("_host", None)
                yield cur
        finally:
            with lock:
                state["stop"] = True
                space.notify_all()
                ready.notify_all()
            for t in threads:
                t.join()
