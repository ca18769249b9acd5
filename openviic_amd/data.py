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


def predict_feature_files(model, vocab, paths: Sequence[str], batch_size: int, beam_size: int = 5, slots: int = 2,
                          keys: Optional[Iterable[str]] = None, trusted: bool = False):
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
    """
    import torch

    from .vocab import captions_from_ids

    device = next(model.parameters()).device
    if device.type != "cuda":
        raise RuntimeError("predict_feature_files needs the model on a HIP device; there is no CPU path")
    keys = tuple(keys) if keys is not None else None
    slots = max(1, int(slots))
    state = getattr(model, "_predict_pipeline", None)           # streams and pinned buffers live as long as the model:
    if state is None or len(state["decode"]) < slots:           # a fresh stream would mean a fresh workspace and graph
        state = model._predict_pipeline = {"copy": torch.cuda.Stream(device=device),
                                           "decode": [torch.cuda.Stream(device=device) for _ in range(slots)],
                                           "pinned": [dict() for _ in range(slots)]}
    copy_stream, decode_streams, pinned = state["copy"], state["decode"], state["pinned"]
    pending = [None] * slots                         # (filenames, pinned ids, done event) of the batch in flight on a slot
    results = []

    def finish(slot):
        entry, pending[slot] = pending[slot], None
        if entry is not None:
            names, ids_host, done = entry
            done.synchronize()
            results.extend(zip(names, captions_from_ids(vocab, ids_host)))

    def pinned_like(slot, name, shape, dtype):
        numel = 1
        for extent in shape:
            numel *= int(extent)
        buf = pinned[slot].get(name)
        if buf is None or buf.numel() < numel or buf.dtype != dtype:
            buf = pinned[slot][name] = torch.empty(max(numel, 1), dtype=dtype).pin_memory()
        return buf[:numel].view(tuple(shape))

    index = 0
    with torch.no_grad():
        for first in range(0, len(paths), batch_size):
            host = batch_from_feature_files(paths[first:first + batch_size], keys=keys, trusted=trusted)
            slot = index % slots
            index += 1
            finish(slot)                               # the slot's pinned buffers are free again once its last batch is done
            items = InstanceList()
            with torch.cuda.stream(copy_stream):
                for name, value in host.items():
                    if isinstance(value, torch.Tensor):
                        stage = pinned_like(slot, name, value.shape, value.dtype)
                        stage.copy_(value)
                        dev = stage.to(device, non_blocking=True)
                        dev.record_stream(decode_streams[slot])
                        items[name] = dev
                    else:
                        items[name] = value
                ready = torch.cuda.Event()
                ready.record(copy_stream)
            with torch.cuda.stream(decode_streams[slot]):
                decode_streams[slot].wait_event(ready)
                outs, _ = model.beam_search(items, batch_size=items.batch_size, beam_size=beam_size, out_size=1)
                ids_host = pinned_like(slot, "__ids__", outs.shape, outs.dtype)
                ids_host.copy_(outs, non_blocking=True)
                done = torch.cuda.Event()
                done.record(decode_streams[slot])
            pending[slot] = (list(host["filename"]) if "filename" in host else [None] * items.batch_size, ids_host, done)
    for step in range(slots):                          # oldest first
        finish((index + step) % slots)
    return results
