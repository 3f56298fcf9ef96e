"""Data path feeding the hot path: `dataset.py:13-190` (Dataset), `:193-272`
(TextDataset), `utils/tools.py:33-110` (to_device), `:334-371` (pad_1D / pad_2D / pad_3D).

Same on-disk format (`<preprocessed_path>/{mel,pitch,energy,duration,phones_per_word,attn_prior}/
<spk>-<kind>-<base>.npy`, `speakers.json`, `train.txt` lines `base|spk|{phones}|raw`), same constructor
arguments, sample dict, 17-/10-slot batch tuples and batch-of-batches collation as the reference, so a batch
from here goes straight into `MixGANTTS.forward(*(batch[2:]))`.  What is different is how the bytes reach
the GPU, designed for one process per GPU:

* `RankShardSampler` -- the reference shuffles one global index list in a single process
  (`train.py:34-39`, `shuffle=True`, no workers); here every rank derives the same seeded permutation per
  epoch and takes every `world`-th *group* (group = `group_size * batch_size` items, the unit
  `collate_fn` sorts inside), so all ranks see the same number of equally sized groups per epoch.
* `PrefetchLoader` -- a background thread does the `np.load`s + collation of the next groups while the
  GPU works on the current one, stages every array of a batch in pinned host memory and issues the
  host->device copies on its own HIP stream; the consumer only waits on an event.  It yields what
  `to_device` would return (same dtypes: `.long()` ids, `.float()` mels / priors).

The text front-end (`text/`: cleaners, symbol table) is out of scope; `text_to_sequence` is injected (by
default the reference's own `text.text_to_sequence` is imported lazily when that package is importable).
"""
import json
import os
import queue
import threading

import numpy as np
import torch


# ------------------------------------------------------------------ padding (utils/tools.py:334-371)
def pad_1D(inputs, PAD=0):
    max_len = max(len(x) for x in inputs)
    out = np.full((len(inputs), max_len), PAD, dtype=np.result_type(*[np.asarray(x).dtype for x in inputs]))
    for i, x in enumerate(inputs):
        out[i, :len(x)] = x
    return out


def pad_2D(inputs, maxlen=None):
    max_len = maxlen if maxlen else max(np.shape(x)[0] for x in inputs)
    for x in inputs:
        if np.shape(x)[0] > max_len:
            raise ValueError("not max_len")
    width = {np.shape(x)[1] for x in inputs}
    if len(width) != 1:
        raise ValueError("all arrays of a batch must share their second dimension, got %s" % sorted(width))
    out = np.zeros((len(inputs), max_len, width.pop()), dtype=np.result_type(*[np.asarray(x).dtype for x in inputs]))
    for i, x in enumerate(inputs):
        out[i, :np.shape(x)[0]] = x
    return out


def pad_3D(inputs, B, T, L):
    out = np.zeros((B, T, L), dtype=np.float32)
    for i, x in enumerate(inputs):
        out[i, :np.shape(x)[0], :np.shape(x)[1]] = x
    return out


def _default_text_to_sequence():
    try:
        from text import text_to_sequence      # the reference's own front-end, when running inside its tree
    except Exception as e:                      # pragma: no cover - depends on the host tree
        raise ImportError("pass text_to_sequence=... (the text front-end is not part of this package): %s" % e)
    return text_to_sequence


def _read_meta(path):
    name, speaker, text, raw = [], [], [], []
    with open(path, "r", encoding="utf-8") as f:
        for line in f.readlines():
            n, s, t, r = line.strip("\n").split("|")
            name.append(n)
            speaker.append(s)
            text.append(t)
            raw.append(r)
    return name, speaker, text, raw


class Dataset(torch.utils.data.Dataset):
    """dataset.py:13-190.  Extra keyword: `text_to_sequence` (callable(text, cleaners) -> ids)."""

    KINDS = ("mel", "pitch", "energy", "duration", "phones_per_word", "attn_prior")

    def __init__(self, filename, args, preprocess_config, model_config, train_config, sort=False, drop_last=False,
                 text_to_sequence=None, mmap=True):
        self.model = args.model
        self.preprocess_config = preprocess_config
        self.dataset_name = preprocess_config["dataset"]
        self.preprocessed_path = preprocess_config["path"]["preprocessed_path"]
        self.cleaners = preprocess_config["preprocessing"]["text"]["text_cleaners"]
        self.batch_size = train_config["optimizer"]["batch_size" if self.model != "shallow" else "batch_size_shallow"]
        self.load_spker_embed = model_config["multi_speaker"] \
            and preprocess_config["preprocessing"]["speaker_embedder"] != "none"
        self.basename, self.speaker, self.text, self.raw_text = self.process_meta(filename)
        with open(os.path.join(self.preprocessed_path, "speakers.json")) as f:
            self.speaker_map = json.load(f)
        self.sort = sort
        self.drop_last = drop_last
        self._t2s = text_to_sequence
        self._mmap = "r" if mmap else None

    def __len__(self):
        return len(self.text)

    def _load(self, kind, speaker, basename):
        path = os.path.join(self.preprocessed_path, kind, "{}-{}-{}.npy".format(speaker, kind, basename))
        return np.load(path, mmap_mode=self._mmap, allow_pickle=False)

    def __getitem__(self, idx):
        if self._t2s is None:
            self._t2s = _default_text_to_sequence()
        basename, speaker = self.basename[idx], self.speaker[idx]
        arrs = {k: self._load(k, speaker, basename) for k in self.KINDS}
        spker_embed = np.load(os.path.join(self.preprocessed_path, "spker_embed",
                                           "{}-spker_embed.npy".format(speaker)),
                              allow_pickle=False) if self.load_spker_embed else None
        return {
            "id": basename,
            "speaker": self.speaker_map[speaker],
            "text": np.array(self._t2s(self.text[idx], self.cleaners)),
            "raw_text": self.raw_text[idx],
            "mel": arrs["mel"],
            "pitch": arrs["pitch"],
            "energy": arrs["energy"],
            "duration": arrs["duration"],
            "word_boundary": arrs["phones_per_word"],
            "spker_embed": spker_embed,
            "attn_prior": arrs["attn_prior"],
        }

    def process_meta(self, filename):
        return _read_meta(os.path.join(self.preprocessed_path, filename))

    def reprocess(self, data, idxs):
        """dataset.py:123-169: the 17-slot batch tuple."""
        pick = lambda k: [data[i][k] for i in idxs]  # noqa: E731
        texts, mels, wbs = pick("text"), pick("mel"), pick("word_boundary")
        spker_embeds = np.concatenate(np.array(pick("spker_embed")), axis=0) if self.load_spker_embed else None
        text_w_lens = np.array([w.shape[0] for w in wbs])
        text_lens = np.array([t.shape[0] for t in texts])
        mel_lens = np.array([m.shape[0] for m in mels])
        return (
            pick("id"),
            pick("raw_text"),
            np.array(pick("speaker")),
            pad_1D(texts),
            text_lens,
            max(text_lens),
            pad_1D(wbs),
            text_w_lens,
            max(text_w_lens),
            spker_embeds,
            pad_3D(pick("attn_prior"), len(idxs), max(text_lens), max(mel_lens)),
            pad_2D(mels),
            mel_lens,
            max(mel_lens),
            pad_1D(pick("pitch")),
            pad_1D(pick("energy")),
            pad_1D(pick("duration")),
        )

    def collate_fn(self, data):
        """dataset.py:171-190: order the group by text length (longest first) when `sort`, cut it into consecutive
        sub-batches of `batch_size`; a short remainder becomes a last sub-batch unless `drop_last`."""
        order = list(range(len(data)))
        if self.sort:
            # np.argsort(-len) semantics: descending, and among equal lengths the order numpy's default
            # (introsort, not stable) produces -- delegate to it so ties break exactly like the reference
            order = np.argsort(-np.array([d["text"].shape[0] for d in data])).tolist()
        bs = self.batch_size
        n_full = len(order) // bs
        cuts = [order[k * bs:(k + 1) * bs] for k in range(n_full)]
        rest = order[n_full * bs:]
        if rest and not self.drop_last:
            cuts.append(rest)
        return [self.reprocess(data, c) for c in cuts]


class TextDataset(torch.utils.data.Dataset):
    """dataset.py:193-272 (synthesis from a text list: no mels)."""

    def __init__(self, filepath, preprocess_config, model_config, text_to_sequence=None):
        self.cleaners = preprocess_config["preprocessing"]["text"]["text_cleaners"]
        self.preprocessed_path = preprocess_config["path"]["preprocessed_path"]
        self.load_spker_embed = model_config["multi_speaker"] \
            and preprocess_config["preprocessing"]["speaker_embedder"] != "none"
        self.basename, self.speaker, self.text, self.raw_text = self.process_meta(filepath)
        with open(os.path.join(self.preprocessed_path, "speakers.json")) as f:
            self.speaker_map = json.load(f)
        self._t2s = text_to_sequence

    def __len__(self):
        return len(self.text)

    def __getitem__(self, idx):
        if self._t2s is None:
            self._t2s = _default_text_to_sequence()
        basename, speaker = self.basename[idx], self.speaker[idx]
        ppw = np.load(os.path.join(self.preprocessed_path, "phones_per_word",
                                   "{}-phones_per_word-{}.npy".format(speaker, basename)), allow_pickle=False)
        spker_embed = np.load(os.path.join(self.preprocessed_path, "spker_embed",
                                           "{}-spker_embed.npy".format(speaker)),
                              allow_pickle=False) if self.load_spker_embed else None
        return (basename, self.speaker_map[speaker], np.array(self._t2s(self.text[idx], self.cleaners)),
                self.raw_text[idx], ppw, spker_embed)

    def process_meta(self, filename):
        return _read_meta(filename)

    def collate_fn(self, data):
        texts = [d[2] for d in data]
        wbs = [d[4] for d in data]
        text_lens = np.array([t.shape[0] for t in texts])
        text_w_lens = np.array([w.shape[0] for w in wbs])
        spker_embeds = np.concatenate(np.array([d[5] for d in data]), axis=0) if self.load_spker_embed else None
        return ([d[0] for d in data], [d[3] for d in data], np.array([d[1] for d in data]), pad_1D(texts), text_lens,
                max(text_lens), pad_1D(wbs), text_w_lens, max(text_w_lens), spker_embeds)


# ------------------------------------------------------------------ host -> device (utils/tools.py:33-110)
# slot -> torch dtype after the reference's .long() / .float() (None: from_numpy as is; "py": left on the host)
_SLOTS17 = ("py", "py", torch.long, torch.long, None, "py", torch.long, None, "py", torch.float32, torch.float32,
            torch.float32, None, "py", torch.float32, None, torch.long)
_SLOTS10 = _SLOTS17[:10]


def _slot_to_device(x, kind, device, pin, non_blocking):
    if kind == "py" or x is None:
        return x
    t = torch.from_numpy(np.ascontiguousarray(x))
    if kind is not None:
        t = t.to(kind)
    if pin:
        t = t.pin_memory()
    return t.to(device, non_blocking=non_blocking)


def to_device(data, device, pin=False, non_blocking=False):
    """utils/tools.py:33-110: numpy batch tuple (17 or 10 slots) -> list / tuple with device tensors."""
    if len(data) == 17:
        return [_slot_to_device(x, k, device, pin, non_blocking) for x, k in zip(data, _SLOTS17)]
    if len(data) == 10:
        return tuple(_slot_to_device(x, k, device, pin, non_blocking) for x, k in zip(data, _SLOTS10))
    raise ValueError("expected a 17- or 10-slot batch, got %d slots" % len(data))


# ------------------------------------------------------------------ per-rank sharding + prefetch
class RankShardSampler:
    """Yields, for this rank, lists of dataset indices -- one list per group of `group_items` items -- from a
    permutation that is identical on every rank (seed + epoch).  Groups are dealt round-robin, the ragged
    remainder (fewer than `world` groups, or a short last group) is dropped so every rank runs the same
    number of steps (a collective in the training step would otherwise hang)."""

    def __init__(self, n_items, group_items, rank=0, world=1, seed=1234, shuffle=True):
        if not (0 <= rank < world):
            raise ValueError("rank %d outside world %d" % (rank, world))
        self.n, self.group, self.rank, self.world, self.seed, self.shuffle = n_items, group_items, rank, world, seed, shuffle
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return (self.n // self.group) // self.world

    def __iter__(self):
        order = np.random.default_rng(self.seed + self.epoch).permutation(self.n) if self.shuffle else np.arange(self.n)
        for g in range(len(self)):
            k = (g * self.world + self.rank) * self.group
            yield order[k:k + self.group].tolist()


_NP_OF = {torch.long: np.int64, torch.float32: np.float32}
_TORCH_OF = {np.dtype(d).str: torch.from_numpy(np.empty(0, dtype=d)).dtype
             for d in (np.int64, np.int32, np.float32, np.float64, np.int16, np.uint8, np.bool_)}


class _PinnedArena:
    """Growable pinned host buffer, reused once the copy that last read it has completed."""

    def __init__(self):
        self.buf, self.event = None, None

    def get(self, nbytes):
        if self.event is not None:
            self.event.synchronize()
            self.event = None
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        return self.buf

    def mark(self, event):
        self.event = event


class PrefetchLoader:
    """Iterates the rank's groups; each item is the list of sub-batches `collate_fn` made of one group, already
    on `device` (what the reference's `for batchs in loader: for batch in batchs: to_device(batch)` sees).

    depth   -- groups in flight ahead of the consumer
    workers -- threads doing np.load + collate.  One is the measured optimum with a warm page cache (a second
               Python thread mostly fights the training thread for the GIL: tests/perf_configs.py data); raise
               it only when the files come from slow storage.
    """

    _DONE = object()

    def __init__(self, dataset, sampler, device, depth=2, workers=1):
        self.ds, self.sampler, self.device = dataset, sampler, torch.device(device)
        self.depth, self.workers = max(1, depth), max(1, workers)
        self.cuda = self.device.type == "cuda"
        # pinned staging (2 per worker) and copy streams live as long as the loader: pinning host memory costs
        # tens of milliseconds per buffer, far more than loading a group
        self._arenas = [[_PinnedArena(), _PinnedArena()] for _ in range(self.workers)]
        self._streams = [None] * self.workers

    def __len__(self):
        return len(self.sampler)

    def _stage(self, idxs, stream, arena):
        batchs = self.ds.collate_fn([self.ds[i] for i in idxs])
        if not self.cuda:
            return [to_device(b, self.device) for b in batchs], None
        # One pinned staging buffer and ONE host->device copy per group: every array of every sub-batch is
        # converted to its final dtype while it is packed (256-byte aligned) into the arena; the device
        # tensors are views into a single device buffer.  (A pin_memory() per tensor costs a hipHostMalloc
        # each -- slower than the synchronous loop it is meant to beat.)
        plan, total = [], 0
        for b in batchs:
            kinds = _SLOTS17 if len(b) == 17 else _SLOTS10
            row = []
            for x, kind in zip(b, kinds):
                if kind == "py" or x is None:
                    row.append(None)
                    continue
                x = np.asarray(x)
                dt = x.dtype if kind is None else _NP_OF[kind]
                row.append((total, x, np.dtype(dt)))
                total += (x.size * np.dtype(dt).itemsize + 255) // 256 * 256
            plan.append(row)
        host = arena.get(max(total, 256))
        hview = host.numpy()
        for row in plan:
            for ent in row:
                if ent is not None:
                    off, x, dt = ent
                    dst = hview[off:off + x.size * dt.itemsize].view(dt).reshape(x.shape)
                    np.copyto(dst, x, casting="unsafe")
        with torch.cuda.stream(stream):
            dev = torch.empty(max(total, 256), dtype=torch.uint8, device=self.device)
            dev.copy_(host[:max(total, 256)], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        arena.mark(ev)
        out = []
        for b, row in zip(batchs, plan):
            slots = []
            for x, ent in zip(b, row):
                if ent is None:
                    slots.append(x)
                else:
                    off, a, dt = ent
                    t = dev[off:off + a.size * dt.itemsize].view(_TORCH_OF[dt.str]).view(a.shape)
                    slots.append(t)
            out.append(slots if len(b) == 17 else tuple(slots))
        return out, ev

    def __iter__(self):
        groups = list(iter(self.sampler))
        slots = [queue.Queue(maxsize=1) for _ in groups]
        gate = threading.Semaphore(self.depth)
        stop = threading.Event()
        cursor = iter(range(len(groups)))
        lock = threading.Lock()

        def work(w):
            if self.cuda and self._streams[w] is None:
                self._streams[w] = torch.cuda.Stream(self.device)
            stream, arenas, turn = self._streams[w], self._arenas[w], 0
            while not stop.is_set():
                gate.acquire()
                with lock:
                    g = next(cursor, None)
                if g is None or stop.is_set():
                    gate.release()
                    return
                try:
                    slots[g].put(self._stage(groups[g], stream, arenas[turn]))
                    turn ^= 1
                except BaseException as e:  # surfaced in the consumer
                    slots[g].put(e)

        threads = [threading.Thread(target=work, args=(w,), daemon=True) for w in range(self.workers)]
        for t in threads:
            t.start()
        try:
            for g in range(len(groups)):
                item = slots[g].get()
                gate.release()
                if isinstance(item, BaseException):
                    raise item
                batchs, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    for b in batchs:      # tensors were allocated on the copy stream
                        for x in b:
                            if torch.is_tensor(x):
                                x.record_stream(torch.cuda.current_stream(self.device))
                yield batchs
        finally:
            stop.set()
            for _ in threads:
                gate.release()
