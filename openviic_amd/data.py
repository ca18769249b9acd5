"""Real-data input path: per-image feature files -> a batch.

The reference stores one ``{image_id}.npy`` per image: ``np.save`` of a ``dict`` with
``region_features [n, d]``, ``region_boxes [n, 4]``, ``grid_features``, ... and reads it back with
``np.load(path, allow_pickle=True)[()]`` (``data_utils/dataset.py:88-92``); the collate step
(``data_utils/utils.py:120-121`` -> ``utils/instance.py:36-55,156-171``) zero-pads ragged region
counts, and those zero rows are what ``FeatureEmbedding`` turns into the padding mask.

A pickled dict can execute code on load, so ``load_feature_file`` refuses pickles unless
``trusted=True`` is passed; ``.npz`` archives of plain arrays load without pickle.
"""
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


def batch_from_feature_files(paths: Sequence[str], keys: Optional[Iterable[str]] = None, trusted: bool = False,
                             device=None) -> InstanceList:
    """Load and collate feature files into an ``InstanceList`` (ragged region counts zero-padded)."""
    keys = tuple(keys) if keys is not None else None
    instances = []
    for path in paths:
        feats = load_feature_file(path, trusted=trusted)
        fields = {k: np.asarray(v, dtype=np.float32) for k, v in feats.items()
                  if (keys is None and k in FEATURE_KEYS) or (keys is not None and k in keys)}
        fields["filename"] = os.path.basename(path)
        instances.append(Instance(**fields))
    batch = InstanceList(instances)
    return batch.to(device) if device is not None else batch
