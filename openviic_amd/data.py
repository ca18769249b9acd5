"""Real-data input path: per-image feature files -> a batch.

The reference stores one ``{image_id}.npy`` per image: ``np.save`` of a ``dict`` with
``region_features [n, d]``, ``region_boxes [n, 4]``, ``grid_features``, ... and reads it back with
``np.load(path, allow_pickle=True)[()]`` (``data_utils/dataset.py:88-92``); the collate step
(``data_utils/utils.py:120-121`` -> ``utils/instance.py:36-55,156-171``) zero-pads ragged region
counts, and those zero rows are what ``FeatureEmbedding`` turns into the padding mask.

A pickled dict can execute code on load, so ``load_feature_file`` refuses pickles unless
``trusted=True`` is passed; ``.npz`` archives of plain arrays load without pickle.
"""
import functools
import os
from typing import Iterable, Optional, Sequence

import numpy as np

from .instance import Instance, InstanceList

FEATURE_KEYS = ("region_features", "region_boxes", "grid_features", "grid_boxes")


def load_feature_file(path: str, trusted: bool = False) -> dict:
    """One image's features as a dict of arrays (reference ``DictionaryDataset.load_features``)."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as archive:
            return {k: archive[k] for k in archive.files}
    if not trusted:
        raise ValueError("{} is a pickled dict (the reference's format); pass trusted=True to unpickle it, "
                         "or store the arrays in an .npz archive".format(path))
    obj = np.load(path, allow_pickle=True)[()]
    if not isinstance(obj, dict):
        raise ValueError("{} does not hold a dict of feature arrays".format(path))
    return obj


def _fields_from_file(path: str, keys, trusted: bool) -> dict:
    """What one image contributes to a batch, as plain numpy arrays (a top-level function: reader PROCESSES import it)."""
    feats = load_feature_file(path, trusted=trusted)
    fields = {k: np.asarray(v, dtype=np.float32) for k, v in feats.items()
              if (keys is None and k in FEATURE_KEYS) or (keys is not None and k in keys)}
    fields["filename"] = os.path.basename(path)
    return fields


def _instance_from_file(path: str, keys, trusted: bool) -> Instance:
    return Instance(**_fields_from_file(path, keys, trusted))


def batch_from_feature_files(paths: Sequence[str], keys: Optional[Iterable[str]] = None, trusted: bool = False,
                             device=None, pool=None) -> InstanceList:
    """Load and collate feature files into an ``InstanceList`` (ragged region counts zero-padded).  ``pool``: an optional
    ``concurrent.futures`` executor that reads the files in parallel (file reads and CRC checks release the GIL); the
    order of the batch is the order of ``paths`` either way."""
    keys = tuple(keys) if keys is not None else None
    if pool is not None:
        # a partial of a top-level function pickles, a lambda does not: works with thread AND process pools
        instances = list(pool.map(functools.partial(_instance_from_file, keys=keys, trusted=trusted), paths))
    else:
        instances = [_instance_from_file(path, keys, trusted) for path in paths]
    batch = InstanceList(instances)
    return batch.to(device) if device is not None else batch


class FeatureFileDataset:
    """Map-style dataset over per-image feature files -- the reference's ``DictionaryDataset`` reduced to what the prediction
    loop reads (``data_utils/dataset.py:74-127``: ``load_features`` per item): item i = the fields of ``paths[i]`` as numpy
    arrays plus ``filename``.  Picklable (a list of paths), so ``torch.utils.data.DataLoader`` workers can hold it."""

    def __init__(self, paths: Sequence[str], keys: Optional[Iterable[str]] = None, trusted: bool = False):
        self.paths = list(paths)
        self.keys = tuple(keys) if keys is not None else None
        self.trusted = trusted

    def __len__(self) -> int:
        return len(self.paths)

    def __getitem__(self, index: int) -> dict:
        return _fields_from_file(self.paths[index], self.keys, self.trusted)


def collate_feature_fields(items: Sequence[dict]) -> dict:
    """The reference's collate (``data_utils/utils.py:120-121`` -> ``InstanceList``: ragged region counts zero-padded) as a
    ``DataLoader`` ``collate_fn``.  Returns a PLAIN dict (name -> tensor or list): a worker hands its tensors to the parent
    through shared memory and the loader's pinning thread walks dicts; ``InstanceList(...)`` is rebuilt by the consumer."""
    return dict(InstanceList([Instance(**fields) for fields in items]))


def _identity(batch):
    return batch


class FeatureBatchDataset:
    """One item = one whole BATCH (``paths[b * batch_size : (b + 1) * batch_size]``), collated by the worker that read its files
    STRAIGHT INTO slot ``b % nslots`` of a ring of shared-memory buffers which the parent has page-locked for the GPU
    (``_SharedPinnedRing``): the worker's collate is the only copy on the host -- the parent receives a slot number and shapes and
    starts the copy to the device.  A field that does not fit its slot (more regions than the ring was sized for), is not
    float32, or has no ring, comes back as an ordinary tensor through the loader's own shared-memory hand-off, zero-padded like
    ``InstanceList`` does it.  Used with ``DataLoader(batch_size=None)``: at most ``prefetch_factor * workers`` batches are being
    written ahead of the one the parent fetched last, which is what the ring's length is chosen against."""

    def __init__(self, paths: Sequence[str], batch_size: int, keys, trusted: bool, ring: dict, nslots: int):
        self.paths, self.batch_size = list(paths), int(batch_size)
        self.keys, self.trusted = (tuple(keys) if keys is not None else None), trusted
        self.ring, self.nslots = ring, int(nslots)

    def __len__(self) -> int:
        return (len(self.paths) + self.batch_size - 1) // self.batch_size

    def __getitem__(self, b: int) -> dict:
        import torch
        from .instance import _pad_rows
        items = [_fields_from_file(path, self.keys, self.trusted) for path in self.paths[b * self.batch_size:(b + 1) * self.batch_size]]
        slot = b % self.nslots
        out = {"filename": [item["filename"] for item in items]}
        for name in items[0]:
            if name == "filename":
                continue
            arrays = [item[name] for item in items]
            first = arrays[0]
            buffers = self.ring.get(name)
            regular = first.ndim >= 1 and all(a.dtype == np.float32 and a.shape[1:] == first.shape[1:] for a in arrays)
            if buffers is not None and regular:
                longest = max(a.shape[0] for a in arrays)
                shape = (len(arrays), longest) + tuple(first.shape[1:])
                numel = int(np.prod(shape))
                if numel <= buffers[slot].numel():
                    dst = buffers[slot][:numel].view(shape).numpy()
                    for i, a in enumerate(arrays):
                        dst[i, :a.shape[0]] = a
                        if a.shape[0] < longest:
                            dst[i, a.shape[0]:] = 0             # the reference's zero rows = padding (utils/instance.py:156-171)
                    out[name] = ("__ring__", slot, shape)
                    continue
            out[name] = _pad_rows([torch.as_tensor(a) for a in arrays])
        return out


class _SharedPinnedRing:
    """``nslots`` float32 buffers per field in SHARED memory (worker processes map them) that this process has registered with the
    HIP runtime as page-locked (``hipHostRegister`` through ``torch.cuda.cudart()``): a worker's collate lands where the copy
    engine reads.  Lives as long as the model's prediction pipeline; registration happens once (it updates the GPU's page
    tables -- not something to do per batch)."""

    def __init__(self, capacities: dict, nslots: int):
        import torch
        self.nslots, self.capacities = int(nslots), dict(capacities)
        self.buffers, self._registered = {}, []
        runtime = torch.cuda.cudart()
        for name, numel in capacities.items():
            self.buffers[name] = []
            for _ in range(self.nslots):
                buf = torch.empty(max(int(numel), 1), dtype=torch.float32).share_memory_()
                err = runtime.cudaHostRegister(buf.data_ptr(), buf.numel() * 4, 0)
                if int(err) != 0:
                    self.release()
                    raise RuntimeError("hipHostRegister failed with code {}".format(int(err)))
                self._registered.append(buf)
                self.buffers[name].append(buf)

    def fits(self, capacities: dict, nslots: int) -> bool:
        return nslots <= self.nslots and all(self.capacities.get(name, -1) >= numel for name, numel in capacities.items())

    def release(self):
        import torch
        runtime = torch.cuda.cudart()
        for buf in self._registered:
            runtime.cudaHostUnregister(buf.data_ptr())
        self._registered, self.buffers = [], {}

    def __del__(self):
        try:
            self.release()
        except Exception:                                  # interpreter shutdown
            pass


def _ring_capacities(paths: Sequence[str], batch_size: int, keys, trusted: bool, headroom: float = 1.25) -> dict:
    """Floats per slot and field, from a sample of the files: batch_size x (longest sampled first dimension x headroom) x the rest."""
    sample = [paths[i] for i in sorted({int(j * (len(paths) - 1) / 15) for j in range(16)})] if len(paths) > 16 else list(paths)
    longest, rest = {}, {}
    for path in sample:
        for name, array in _fields_from_file(path, keys, trusted).items():
            if name == "filename" or array.ndim < 1 or array.dtype != np.float32:
                continue
            longest[name] = max(longest.get(name, 0), array.shape[0])
            rest[name] = int(np.prod(array.shape[1:])) if array.ndim > 1 else 1
    return {name: int(batch_size * (int(longest[name] * headroom) + 1) * rest[name]) for name in longest}


def _worker_context(context):
    """``"forkserver"`` -> a forkserver context with torch pre-loaded.  Workers are then forked from a small server process that
    has torch and this module imported already: they come up in milliseconds (spawned interpreters import torch one after the
    other -- the parent blocks on each worker's start-up pipe -- 0.9 s per worker), and nothing is ever fork()ed from the caller's
    own (HIP-initialised) process."""
    if context == "forkserver":
        import multiprocessing
        context = multiprocessing.get_context("forkserver")
        context.set_forkserver_preload(["torch", "numpy", __name__])
    return context


def feature_file_loader(paths: Sequence[str], batch_size: int, workers: int, keys: Optional[Iterable[str]] = None,
                        trusted: bool = False, pin_memory: bool = False, prefetch_factor: int = 2, context=None):
    """``DataLoader`` over feature files the way the reference feeds its loops (``trainers/base_trainer.py:40-80``: worker
    processes, ``collate_fn``): consecutive groups of ``batch_size`` paths, in order, each batch read and collated by ONE worker
    process and handed over through shared memory (a 105 MB batch at B = 256 is not pickled through a pipe: that is what
    capped round 3's reader pool at 350 MB/s).  ``pin_memory`` (the loader's own pinning thread) is OFF by default:
    ``predict_feature_files`` copies into pinned buffers it allocates ONCE -- the pinning thread allocates and frees 105 MB of
    pinned memory per batch, and every such allocation updates the GPU's page tables under the running decode (measured: the
    decode of a batch then takes 183 ms instead of 11)."""
    from torch.utils.data import DataLoader
    context = _worker_context(context) if workers > 0 else None
    return DataLoader(FeatureFileDataset(paths, keys, trusted), batch_size=batch_size, shuffle=False, num_workers=workers,
                      collate_fn=collate_feature_fields, pin_memory=pin_memory and workers > 0,
                      prefetch_factor=prefetch_factor if workers > 0 else None,
                      multiprocessing_context=context if workers > 0 else None, persistent_workers=False)


def predict_feature_files(model, vocab, paths: Sequence[str], batch_size: int, beam_size: int = 5, slots: Optional[int] = None,
                          keys: Optional[Iterable[str]] = None, trusted: bool = False, workers: int = 0,
                          loader_context="forkserver", early_exit: bool = False, direct: bool = True):
    """The reference's prediction loop (``trainers/vi_trainer.py:241-252``: per batch ``items.to(device)`` ->
    ``model.beam_search(items, batch_size, beam_size, out_size=1)`` -> ``decode_caption`` -> duplicate collapse) as a
    software pipeline on ONE host thread:

    * batch i is read and collated on the host while the GPU still decodes batch i - 1 (``slots`` batches in flight);
    * its tensors go into the slot's PINNED staging buffers (allocated once) and cross PCIe on a copy stream
      (``non_blocking``); an event hands them to the slot's decode stream, which runs ``beam_search`` (own engine
      workspace, hipGraph replay) and copies the token ids back into pinned memory;
    * the strings of a batch are built when its slot comes round again (or at the end).

    Why staging matters even at batch size 1: a ``.to(device)`` from pageable memory makes the HIP runtime lock and unlock
    the pages around the copy, and the decode that follows it then takes 16 ms instead of 5 ms (measured, DESIGN.md section
    5b: GPU page-table updates); pinned buffers are locked once.  No loader thread: parsing the reference's feature files
    is Python work, and a second Python thread only makes the launching thread queue for the GIL (measured: 110 ms per
    image instead of 46 ms).  Batches are the reference's: consecutive groups of ``batch_size`` paths (its loaders use 1
    for the test set and ``DICT_BATCH_SIZE // beam`` for validation, ``trainers/base_trainer.py:63-80``).  Returns
    ``[(filename, caption)]`` in input order; results are identical to the sequential loop
    (``tests/test_engine_gpu.py::test_pipelined_prediction_matches_the_sequential_loop``).

    ``workers > 0`` (round 4): the host side of the loop -- parsing the feature files and collating a batch, Python work that
    one thread does at 600-880 images/s -- moves into ``workers`` ``DataLoader`` processes (``feature_file_loader``); batches
    come back through shared memory, a copier thread moves each into a ring of pinned buffers that live as long as the model,
    and the launching thread only ever sees pinned tensors.  Same strings, in the same order.  ``loader_context``: the
    ``multiprocessing`` start method of the workers.  The default is ``"forkserver"`` (workers forked from a small server
    process that has torch imported: up in milliseconds; ``"spawn"`` works too, 0.9 s per worker), NOT the platform's fork:
    while a fork()ed child of this HIP-initialised process is alive, every page the parent writes is
    copied first (copy-on-write, GPU-mapped host memory included) and the launching thread takes 64 ms per batch instead of 12
    (tools/loader_stall_probe.py; a pure-Python thread holding the GIL costs about as much -- the copier thread's Python work
    is a few calls per batch).

    ``direct=True`` (default, with workers): the workers collate each batch STRAIGHT INTO a ring of shared-memory buffers that this
    process has page-locked for the GPU (``FeatureBatchDataset`` / ``_SharedPinnedRing``) -- no copier thread, no second copy on the
    host: the launching thread receives a slot number and starts the copy to the device.  The ring has
    ``2 * workers + slots + 2`` slots of the batch's size (2.3 GB at B = 256 with 8 workers), allocated and registered once per
    model.  ``direct=False`` keeps the copier thread of the first version (a collated batch comes back through the loader's own
    shared memory and is copied into pinned buffers: 8 ms of a 22 ms batch period at B = 256).

    ``early_exit=True``: decode with ``ovc_beam_search_early`` -- no step is issued once every beam of the batch has ended (real
    captions end well before ``max_len``); same strings.  That call blocks the launching thread until its batch is one step
    from done, so each slot's search runs on its own host thread (the call is one C function: the GIL is released for all of it)
    and the slots keep overlapping.

    ``slots``: batches in flight, each on its own decode stream.  Default 4: a small batch is a chain of ~730 dependent launches
    of a few workgroups each -- four of them overlap almost freely (B = 1, files -> strings, 8 workers: 430 captions/s with two
    streams, 725 with four) --, a batch of 256 fills the chip better with four than with two (19.0k against 16.8k captions/s
    with the shared ring), and a fifth search serialises with the others whatever GPU_MAX_HW_QUEUES is (270-390 at B = 1).
    """
    import sys
    import time

    import torch

    from .vocab import captions_from_ids

    trace = os.environ.get("OVC_PREDICT_TRACE", "0") != "0"        # per-batch phase times on stderr (tools/loader_probe.py)
    device = next(model.parameters()).device
    if device.type != "cuda":
        raise RuntimeError("predict_feature_files needs the model on a HIP device; there is no CPU path")
    keys = tuple(keys) if keys is not None else None
    slots = 4 if slots is None else max(1, int(slots))
    state = getattr(model, "_predict_pipeline", None)           # streams and pinned buffers live as long as the model:
    if state is None or len(state["decode"]) < slots:           # a fresh stream would mean a fresh workspace and graph
        state = model._predict_pipeline = {"copy": torch.cuda.Stream(device=device),
                                           "decode": [torch.cuda.Stream(device=device) for _ in range(slots)],
                                           "pinned": [dict() for _ in range(slots)]}
    copy_stream, decode_streams, pinned = state["copy"], state["decode"], state["pinned"]
    pending = [None] * slots                         # (filenames, pinned ids, done event) of the batch in flight on a slot
    results = []

    def search(slot, items, ready, names):
        with torch.no_grad(), torch.cuda.stream(decode_streams[slot]):
            decode_streams[slot].wait_event(ready)
            outs, _ = model.beam_search(items, batch_size=items.batch_size, beam_size=beam_size, out_size=1, early_exit=early_exit)
            ids_host = pinned_like(slot, "__ids__", outs.shape, outs.dtype)
            ids_host.copy_(outs, non_blocking=True)
            done = torch.cuda.Event()
            done.record(decode_streams[slot])
        return names, ids_host, done

    searchers = None
    if early_exit and slots > 1:
        from concurrent.futures import ThreadPoolExecutor
        searchers = ThreadPoolExecutor(max_workers=slots, thread_name_prefix="ovc-search")

    def finish(slot):
        entry, pending[slot] = pending[slot], None
        if entry is not None:
            names, ids_host, done = entry.result() if searchers else entry
            done.synchronize()
            results.extend(zip(names, captions_from_ids(vocab, ids_host)))
            if ring_of.get(slot) is not None:          # the batch's copy to the device is long done: its ring entry is free again
                free_ring.put(ring_of.pop(slot))

    def pinned_like(slot, name, shape, dtype):
        numel = 1
        for extent in shape:
            numel *= int(extent)
        buf = pinned[slot].get(name)
        if buf is None or buf.numel() < numel or buf.dtype != dtype:
            buf = pinned[slot][name] = torch.empty(max(numel, 1), dtype=dtype).pin_memory()
        return buf[:numel].view(tuple(shape))

    ring_of = {}                                     # decode slot -> staging-ring entry of the batch in flight on it
    free_ring = None
    source = None
    if workers > 0 and direct:
        # The workers write each collated batch into a slot of a shared, page-locked ring; batch b uses slot b % nslots.  The
        # loader keeps at most 2 * workers batches in flight ahead of the one fetched last, and a slot's previous batch (nslots
        # earlier) finished decoding before the batch 2 * workers + 1 after it is fetched (finish() below): nothing is overwritten
        # while the copy engine may still read it.
        try:
            from torch.utils.data import DataLoader
            prefetch = 2
            nslots = prefetch * workers + slots + 2
            capacities = _ring_capacities(paths, batch_size, keys, trusted)
            shared = state.get("shared_ring")
            if shared is None or not shared.fits(capacities, nslots):
                if shared is not None:
                    torch.cuda.synchronize(device)
                    shared.release()
                shared = state["shared_ring"] = _SharedPinnedRing(capacities, nslots)
            batches = DataLoader(FeatureBatchDataset(paths, batch_size, keys, trusted, shared.buffers, shared.nslots), batch_size=None,
                                 shuffle=False, num_workers=workers, prefetch_factor=prefetch, collate_fn=_identity,
                                 multiprocessing_context=_worker_context(loader_context), persistent_workers=False)

            def ring_source():
                for fields in batches:
                    host = InstanceList()
                    for name, value in fields.items():
                        if isinstance(value, (tuple, list)) and len(value) == 3 and value[0] == "__ring__":
                            _, ring_slot, shape = value
                            numel = 1
                            for extent in shape:
                                numel *= int(extent)
                            host[name] = shared.buffers[name][ring_slot][:numel].view(tuple(shape))
                        else:
                            host[name] = value
                    yield None, host
            source = ring_source()
        except (RuntimeError, AttributeError) as error:      # no hipHostRegister in this build of torch, or it refused: the copier path
            print("[predict] shared page-locked ring unavailable ({}); using the copier thread".format(error), file=sys.stderr)
            source = None
    if workers > 0 and source is None:
        # The host side in worker processes: a copier thread takes each collated batch out of the loader's shared memory and
        # into a ring of pinned buffers that live as long as the model (one plain memcpy, GIL released), the launching thread
        # only ever sees pinned tensors.  Ring entries return to the copier when their batch's strings have been built.
        import queue
        import threading
        ring = state.setdefault("ring", [])
        while len(ring) < slots + 2:
            ring.append(dict())
        free_ring, staged_q = queue.Queue(), queue.Queue(maxsize=len(ring))
        for entry in range(len(ring)):
            free_ring.put(entry)
        loader = feature_file_loader(paths, batch_size, workers, keys=keys, trusted=trusted, context=loader_context)
        failure = []
        stopping = threading.Event()                 # set when the launching thread leaves early (an exception): see `finally`

        def stage_batches():
            try:
                it = iter(loader)
                while True:
                    t_a = time.perf_counter()
                    try:
                        fields = next(it)
                    except StopIteration:
                        break
                    t_b = time.perf_counter()
                    entry = free_ring.get()
                    if stopping.is_set():
                        break
                    t_c = time.perf_counter()
                    staged = InstanceList()
                    for name, value in fields.items():
                        if isinstance(value, torch.Tensor):
                            numel = value.numel()
                            buf = ring[entry].get(name)
                            if buf is None or buf.numel() < numel or buf.dtype != value.dtype:
                                buf = ring[entry][name] = torch.empty(max(numel, 1), dtype=value.dtype).pin_memory()
                            stage = buf[:numel].view(value.shape)
                            np.copyto(stage.numpy(), value.contiguous().numpy())
                            staged[name] = stage
                        else:
                            staged[name] = value
                    if trace:
                        print("[predict] staging: loader %.1f ms, free entry %.1f ms, copy %.1f ms"
                              % (1e3 * (t_b - t_a), 1e3 * (t_c - t_b), 1e3 * (time.perf_counter() - t_c)), file=sys.stderr, flush=True)
                    staged_q.put((entry, staged))
            except BaseException as error:            # surfaced in the launching thread
                failure.append(error)
            finally:
                del it                                 # shuts the loader's worker processes down
                staged_q.put(None)                     # never blocks: at most len(ring) - 1 batches are staged at a time
        copier = threading.Thread(target=stage_batches, name="ovc-feature-staging", daemon=True)
        copier.start()

        def staged_source():
            while True:
                item = staged_q.get()
                if item is None:
                    if failure:
                        raise failure[0]
                    return
                yield item
        source = staged_source()
    elif source is None:
        source = ((None, batch_from_feature_files(paths[first:first + batch_size], keys=keys, trusted=trusted))
                  for first in range(0, len(paths), batch_size))
    index = 0
    t_prev = time.perf_counter()
    try:
        with torch.no_grad():
            for entry, host in source:
                t_got = time.perf_counter()
                slot = index % slots
                index += 1
                finish(slot)                               # the slot's pinned buffers are free again once its last batch is done
                t_fin = time.perf_counter()
                items = InstanceList()
                with torch.cuda.stream(copy_stream):
                    for name, value in host.items():
                        if isinstance(value, torch.Tensor):
                            if entry is not None or value.is_pinned():     # already in page-locked memory (the staging ring / the shared ring)
                                stage = value
                            else:
                                stage = pinned_like(slot, name, value.shape, value.dtype)
                                stage.copy_(value)
                            dev = stage.to(device, non_blocking=True)
                            dev.record_stream(decode_streams[slot])
                            items[name] = dev
                        else:
                            items[name] = value
                    ready = torch.cuda.Event()
                    ready.record(copy_stream)
                names = list(host["filename"]) if "filename" in host else [None] * items.batch_size
                # early exit: the search call blocks until its batch is nearly done, so each slot's search runs on its own
                # thread (the call is ONE C function: the GIL is released for all of it) and the slots overlap as before
                pending[slot] = searchers.submit(search, slot, items, ready, names) if searchers else search(slot, items, ready, names)
                ring_of[slot] = entry
                if trace:
                    now = time.perf_counter()
                    print("[predict] batch %d: waited %.1f ms for it, finish(previous on slot) %.1f ms, launch %.1f ms"
                          % (index - 1, 1e3 * (t_got - t_prev), 1e3 * (t_fin - t_got), 1e3 * (now - t_fin)), file=sys.stderr, flush=True)
                    t_prev = now
        for step in range(slots):                      # oldest first
            finish((index + step) % slots)
    finally:
        if searchers:
            searchers.shutdown(wait=True)
        if free_ring is not None:                      # leaving early (an exception above): let the copier thread and its loader go
            stopping.set()
            free_ring.put(0)
            while copier.is_alive():
                try:
                    staged_q.get(timeout=0.05)
                except queue.Empty:
                    pass
    return results
