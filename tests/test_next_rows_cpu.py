"""SURVEY.md section 8f "next" rows that need no GPU: ids -> captions, and reference checkpoints."""
import json
import os

import numpy as np
import torch

from helpers import GOLDEN, TINY, TINY_SHAPE, tiny_case
from openviic_amd.builders import build_model
from openviic_amd.checkpoint import load_reference_checkpoint
from openviic_amd.config import model_config
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features
from openviic_amd.vocab import WordVocab, captions_from_ids, collapse_repeated_words
from oracle.captioner import OracleCaptioner


def test_decode_caption_matches_reference_strings():
    g = json.load(open(os.path.join(GOLDEN, "g6_decode_caption.json")))
    vocab = WordVocab(g["itos"], max_caption_length=8)
    assert (vocab.padding_idx, vocab.bos_idx, vocab.eos_idx, vocab.unk_idx, len(vocab)) == (0, 1, 2, 3, 53)
    ids = torch.tensor(g["ids"])
    assert vocab.decode_caption(ids, join_words=True) == g["joined"]
    assert vocab.decode_caption(ids, join_words=False) == g["split"]
    assert captions_from_ids(vocab, ids) == g["collapsed"]
    assert collapse_repeated_words(["a", "a", "b", "a"]) == "a b a"
    enc = vocab.encode_caption(["w00", "nope"])
    assert enc.tolist() == [1, 4, 3, 2, 0, 0, 0, 0]


def test_reference_checkpoint_loads_key_for_key_and_oracle_reproduces_its_output():
    for variant in ("standard_transformer", "meshed_memory_transformer"):
        path = os.path.join(GOLDEN, "g7_reference_checkpoint_%s.pth" % variant)
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        vocab = SyntheticVocab(TINY_SHAPE["V"], TINY_SHAPE["T"])
        cfg = model_config(variant, device="cpu", **TINY)
        model = build_model(cfg, vocab)
        result = load_reference_checkpoint(model, path, strict=True)          # every key, every shape
        assert not result.missing_keys and not result.unexpected_keys
        for k, v in ckpt["state_dict"].items():
            assert torch.equal(model.state_dict()[k], v), k
        feats = synthetic_features(TINY_SHAPE["B"], TINY_SHAPE["N"], TINY["d_feature"], seed=3, ragged=True)
        ids, logp = OracleCaptioner(cfg, ckpt["state_dict"], len(vocab), vocab.max_caption_length).beam_search(feats, TINY_SHAPE["k"])
        assert torch.equal(ids, ckpt["beam_ids"])
        np.testing.assert_allclose(logp.numpy(), ckpt["beam_logp"].numpy(), rtol=1e-5, atol=5e-6)


def test_feature_files_collate_like_the_reference(tmp_path):
    """{image_id}.npy dicts (reference format) and .npz archives -> zero-padded batch."""
    import pytest
    from openviic_amd.data import batch_from_feature_files, load_feature_file
    rng = np.random.default_rng(0)
    paths = []
    for i, n in enumerate((5, 3, 7)):
        feats = {"region_features": rng.standard_normal((n, 8)).astype(np.float32),
                 "region_boxes": rng.random((n, 4)).astype(np.float32), "unused_field": np.zeros(2)}
        path = str(tmp_path / ("%d.npy" % i))
        np.save(path, feats)                                   # what the reference's feature extractor writes
        paths.append(path)
    with pytest.raises(ValueError):
        load_feature_file(paths[0])                            # pickled dict: must be opted into
    batch = batch_from_feature_files(paths, trusted=True)
    assert tuple(batch.region_features.shape) == (3, 7, 8) and tuple(batch.region_boxes.shape) == (3, 7, 4)
    assert batch.region_features[1, 3:].abs().sum() == 0 and batch.region_features[1, :3].abs().sum() > 0
    assert batch.filename == ["0.npy", "1.npy", "2.npy"] and batch.unused_field is None
    mask = batch.region_features.sum(-1) == 0                  # models/utils.py:60: zero rows are padding
    assert mask.sum(1).tolist() == [2, 4, 0]
    npz = str(tmp_path / "9.npz")
    np.savez(npz, region_features=np.ones((2, 8), np.float32))
    assert load_feature_file(npz)["region_features"].shape == (2, 8)


def _g9_instances():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(GOLDEN, "make_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                      # imports only openviic_amd; the reference is touched in main()
    return mod.g9_instances()


def test_collate_matches_the_reference_instance_list(tmp_path):
    """G9 (f2 pinned): the same ragged per-image dicts through the reference's ``InstanceList`` (fixture) and through
    ``openviic_amd.instance`` / ``openviic_amd.data``: identical tensors (values, shapes, dtypes incl. the reference's
    float32-filler promotion), identical non-tensor fields, identical ``batch_size``."""
    from openviic_amd.data import batch_from_feature_files
    from openviic_amd.instance import Instance, InstanceList
    g = np.load(os.path.join(GOLDEN, "g9_collated_batch.npz"))
    lists = json.load(open(os.path.join(GOLDEN, "g9_prediction_loop.json")))["lists"]
    samples = _g9_instances()
    batch = InstanceList([Instance(**d) for d in samples])
    assert batch.batch_size == 4 and batch.get_fields() == list(samples[0].keys())
    for key in ("region_features", "region_boxes", "grid_features", "region_scores", "region_labels"):
        assert str(batch[key].dtype) == str(g[key + "_dtype"]), key
        np.testing.assert_array_equal(batch[key].numpy(), g[key], err_msg=key)
    assert tuple(batch.region_features.shape) == (4, 7, TINY["d_feature"]) and batch.region_labels.dtype == torch.float32
    for key, want in lists.items():
        assert batch[key] == want, key
    assert batch.missing is None and batch.to("cpu").region_boxes.shape == (4, 7, 4)
    # the same images from {image_id}.npy files, the reference's on-disk format (data_utils/dataset.py:88-92)
    paths = []
    for d in samples:
        path = str(tmp_path / ("%d.npy" % d["image_id"]))
        np.save(path, {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items() if k.endswith(("features", "boxes"))})
        paths.append(path)
    from_files = batch_from_feature_files(paths, trusted=True)
    for key in ("region_features", "region_boxes", "grid_features"):
        np.testing.assert_array_equal(from_files[key].numpy(), g[key], err_msg=key)
    # ... and the oracle decodes that batch to the ids / strings the reference's prediction loop produced
    want = json.load(open(os.path.join(GOLDEN, "g9_prediction_loop.json")))
    ckpt = torch.load(os.path.join(GOLDEN, "g7_reference_checkpoint_standard_transformer.pth"), map_location="cpu", weights_only=True)
    cfg = model_config("standard_transformer", device="cpu", **TINY)
    ids, _ = OracleCaptioner(cfg, ckpt["state_dict"], TINY_SHAPE["V"], TINY_SHAPE["T"]).beam_search(from_files.region_features, want["beam_size"])
    np.testing.assert_array_equal(ids.numpy(), g["beam_ids"])
    assert captions_from_ids(WordVocab(want["itos"], TINY_SHAPE["T"]), ids) == want["gens"]


# ---- dual-collaborative encoder: host-side logic (no device work) --------------------------------------------

def test_grid_visibility_mask_matches_reference_cell_lookup():
    """The vectorised mask of the product against the reference's per-box loops (G8 vectors: coordinates one ulp
    either side of the cell edges, inverted boxes, boxes below the first edge, the tiny encoder case)."""
    import numpy as np
    from helpers import golden
    from openviic_amd.modules import grid_visibility_mask
    g = golden("g8_dlct_encoder.npz")
    np.testing.assert_array_equal(grid_visibility_mask(torch.from_numpy(g["edge_boxes"]), 10).numpy(), g["edge_mask_g10"])
    n = g["region_boxes"].shape[1]
    r2g = grid_visibility_mask(torch.from_numpy(g["region_boxes"]), 3).numpy()
    np.testing.assert_array_equal(r2g, g["region2all_mask"][..., n:])
    np.testing.assert_array_equal(np.swapaxes(r2g, 2, 3), g["grid2all_mask"][..., :n])


def test_dual_collaborative_modules_register_and_match_reference_state_dict_surface():
    from openviic_amd.builders import META_ENCODER, META_VISION_EMBEDDING, build_encoder, build_vision_embedding
    from openviic_amd.config import dual_collaborative_config
    emb_cfg, enc_cfg = dual_collaborative_config(d_region=32, d_grid=24, d_model=64, heads=4, d_kv=16, d_ff=128, layers=2)
    for name in ("DualFeatureEmbedding", "GeometricDualFeatureEmbedding"):
        assert META_VISION_EMBEDDING.get(name).__name__ == name
    assert META_ENCODER.get("DualCollaborativeLevelEncoder").__name__ == "DualCollaborativeLevelEncoder"
    emb, enc = build_vision_embedding(emb_cfg), build_encoder(enc_cfg)
    assert sorted(emb.state_dict()) == ["grid_proj.bias", "grid_proj.weight", "region_proj.bias", "region_proj.weight"]
    keys = set(enc.state_dict())
    # key surface of the reference class (encoders.py:116-144): 4 layer stacks, two stream norms, per-head fc_gs
    for stack in ("layers_region", "layers_grid", "region2grid", "grid2region"):
        for i in range(2):
            for leaf in ("mhatt.attention.fc_q.weight", "mhatt.layer_norm.bias", "pwff.fc1.weight", "pwff.fc2.bias",
                         "pwff.layer_norm.weight"):
                assert "%s.%d.%s" % (stack, i, leaf) in keys
    assert {"layer_norm_region.weight", "layer_norm_grid.bias", "fc_gs.3.weight", "fc_gs.0.bias"} <= keys
    assert len(keys) == 4 + 2 * 4 + 4 * 2 * 16
    assert enc.state_dict()["fc_gs.0.weight"].shape == (1, 4)


def test_feature_file_loader_collates_like_the_sequential_loop(tmp_path):
    """SURVEY.md section 8f rank 2 / VERDICT r3 item 8: the DataLoader over feature files (worker processes, the reference's way
    of feeding its loops: trainers/base_trainer.py:40-80) must hand over exactly the batches the sequential collate builds --
    same order, same zero padding of ragged region counts, same non-tensor fields -- including the ragged last batch."""
    import numpy as np
    import torch
    from openviic_amd.data import batch_from_feature_files, collate_feature_fields, feature_file_loader, FeatureFileDataset
    g = torch.Generator().manual_seed(1)
    paths = []
    for i in range(11):
        n = int(torch.randint(3, 8, (1,), generator=g))
        path = str(tmp_path / ("img_%02d.npz" % i))
        np.savez(path, region_features=torch.randn(n, 32, generator=g).numpy(), region_boxes=torch.rand(n, 4, generator=g).numpy())
        paths.append(path)
    assert len(FeatureFileDataset(paths)) == 11 and sorted(FeatureFileDataset(paths, keys=("region_boxes",))[3]) == ["filename", "region_boxes"]
    for workers in (0, 2):
        got = list(feature_file_loader(paths, 4, workers, pin_memory=False))
        assert len(got) == 3
        for i, fields in enumerate(got):
            want = batch_from_feature_files(paths[4 * i:4 * i + 4])
            assert isinstance(fields, dict) and list(fields) == list(want)
            for name in want:
                if isinstance(want[name], torch.Tensor):
                    assert torch.equal(fields[name], want[name]), (workers, i, name)
                else:
                    assert fields[name] == want[name]
    single = collate_feature_fields([FeatureFileDataset(paths)[0]])
    assert single["region_features"].shape[0] == 1 and single["filename"] == ["img_00.npz"]


def test_workers_collate_a_batch_straight_into_a_shared_ring(tmp_path):
    """The zero-copy hand-off of the prediction loop (data.FeatureBatchDataset): one item = one whole batch, written by the worker
    into slot b % nslots of shared-memory buffers (page-locked by the parent on the GPU box; plain shared memory here).  What
    lands in a slot must be exactly the sequential collate -- zero padding of ragged region counts included, over whatever an
    earlier batch left in the slot -- and a batch that does not fit its slot must come back as an ordinary tensor."""
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    from openviic_amd.data import FeatureBatchDataset, _identity, _ring_capacities, batch_from_feature_files
    g = torch.Generator().manual_seed(3)
    paths = []
    for i in range(23):
        n = int(torch.randint(3, 9, (1,), generator=g))
        path = str(tmp_path / ("img_%02d.npz" % i))
        np.savez(path, region_features=torch.randn(n, 16, generator=g).numpy(), region_boxes=torch.rand(n, 4, generator=g).numpy())
        paths.append(path)
    caps = _ring_capacities(paths, 4, None, False)
    assert caps["region_features"] >= 4 * 8 * 16 and caps["region_boxes"] >= 4 * 8 * 4
    nslots = 3
    ring = {"region_features": [torch.full((caps["region_features"],), float("nan")).share_memory_() for _ in range(nslots)],
            "region_boxes": [torch.full((4 * 5 * 4,), float("nan")).share_memory_() for _ in range(nslots)]}     # too small for 6+ regions
    dataset = FeatureBatchDataset(paths, 4, None, False, ring, nslots)
    assert len(dataset) == 6
    for workers in (0, 2):
        through_ring = fallbacks = 0
        for b, fields in enumerate(DataLoader(dataset, batch_size=None, shuffle=False, num_workers=workers, collate_fn=_identity,
                                              prefetch_factor=1 if workers else None)):
            want = batch_from_feature_files(paths[4 * b:4 * b + 4])
            assert fields["filename"] == want["filename"]
            for name in ("region_features", "region_boxes"):
                value = fields[name]
                if isinstance(value, (tuple, list)):
                    tag, slot, shape = value
                    assert tag == "__ring__" and slot == b % nslots
                    numel = int(np.prod(shape))
                    got = ring[name][slot][:numel].view(tuple(shape))
                    through_ring += 1
                else:
                    got = value
                    fallbacks += 1
                assert torch.equal(got, want[name]), (workers, b, name)
        assert through_ring >= 6 and fallbacks >= 1            # the features always fit; some box batches have more than 5 regions
