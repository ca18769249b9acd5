"""Host-side logic that needs no GPU: configuration, registries, containers, the state-dict surface
and the C-ABI library (it must load and export every symbol declared in include/ovc.h; no compute
call is made here)."""
import os
import re

import numpy as np
import pytest
import torch

from openviic_amd import native
from openviic_amd.builders import (META_ARCHITECTURE, META_ATTENTION, META_DECODER, META_ENCODER,
                                   META_TEXT_EMBEDDING, META_VISION_EMBEDDING, Registry, build_model)
from openviic_amd.config import ConfigNode, get_config, model_config
from openviic_amd.instance import Instance, InstanceList
from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_state_dict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_CONFIGS = "/root/reference/configs"


def test_registry_contract():
    reg = Registry("T")

    @reg.register()
    class A:
        pass

    class B:
        pass
    reg.register(B)
    assert reg.get("A") is A and reg.get("B") is B and "A" in reg and dict(reg)["B"] is B
    with pytest.raises(KeyError):
        reg.get("missing")
    with pytest.raises(AssertionError):
        reg.register(B)


def test_registered_names_match_the_reference():
    assert {"StandardTransformerUsingRegion", "StandardTransformerUsingGrid", "MeshedMemoryTransformer",
            "ObjectRelationTransformer"} <= set(META_ARCHITECTURE.names())
    assert {"Encoder", "MultilevelEncoder", "GeometricEncoder"} <= set(META_ENCODER.names())
    assert {"Decoder", "MeshedDecoder"} <= set(META_DECODER.names())
    assert {"ScaledDotProductAttention", "AugmentedMemoryScaledDotProductAttention",
            "AugmentedGeometryScaledDotProductAttention"} <= set(META_ATTENTION.names())
    assert "FeatureEmbedding" in META_VISION_EMBEDDING and "UsualEmbedding" in META_TEXT_EMBEDDING


def test_config_node_semantics():
    cfg = ConfigNode({"MODEL": {"DEVICE": "cuda", "ENCODER": {"LAYERS": 3}}})
    assert cfg.MODEL.ENCODER.LAYERS == 3
    with pytest.raises(AttributeError):
        cfg.MODEL.NOPE
    cfg.merge_from_list([("MODEL.DEVICE", "cpu"), ("MODEL.ENCODER.LAYERS", 2)])
    assert cfg.MODEL.DEVICE == "cpu" and cfg.clone().MODEL.ENCODER.LAYERS == 2
    with pytest.raises(KeyError):
        cfg.merge_from_list([("MODEL.MISSING.X", 1)])


@pytest.mark.parametrize("yaml_name,variant", [
    ("standard_transformer.yaml", "standard_transformer"),
    ("standard_transformer_using_region.yaml", "standard_transformer_using_region"),
    ("meshed_memory_transformer.yaml", "meshed_memory_transformer"),
    ("object_relation_transformer.yaml", "object_relation_transformer"),
    ("attention_on_attention.yaml", "attention_on_attention"),
])
def test_reference_yaml_drops_in_unchanged(yaml_name, variant):
    """The reference's own yaml builds the same model as the programmatic config (only on machines
    where the reference checkout is mounted)."""
    path = os.path.join(REFERENCE_CONFIGS, yaml_name)
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    cfg = get_config(path, {"MODEL.DEVICE": "cpu", "MODEL.VISION_EMBEDDING.D_FEATURE": 2048})
    vocab = SyntheticVocab()
    from_yaml = build_model(cfg.MODEL, vocab).state_dict()
    programmatic = build_model(model_config(variant, device="cpu"), vocab).state_dict()
    assert {k: tuple(v.shape) for k, v in from_yaml.items()} == {k: tuple(v.shape) for k, v in programmatic.items()}


@pytest.mark.parametrize("variant,count", [("standard_transformer", 141), ("meshed_memory_transformer", 165),
                                           ("object_relation_transformer", 157)])
def test_state_dict_surface(variant, count):
    """Checkpoint surface of SURVEY.md section 8b: key count and a few load-bearing shapes."""
    sd = build_model(model_config(variant, device="cpu"), SyntheticVocab()).state_dict()
    assert len(sd) == count
    assert tuple(sd["vision_embedding.proj.weight"].shape) == (512, 2048)
    assert tuple(sd["decoder.fc.weight"].shape) == (10201, 512) and "decoder.fc.bias" not in sd
    assert tuple(sd["decoder.pos_emb.weight"].shape) == (21, 512)
    assert tuple(sd["decoder.running_mask_self_attention"].shape) == (1, 1, 0)
    assert tuple(sd["decoder.layers.0.self_attn.running_keys"].shape) == (0, 512)
    if variant == "meshed_memory_transformer":
        assert tuple(sd["encoder.layers.0.mhatt.attention.m_k"].shape) == (1, 40, 512)
        assert tuple(sd["decoder.layers.2.fc_alphas.1.weight"].shape) == (512, 1024)
    if variant == "object_relation_transformer":
        assert tuple(sd["encoder.fc_gs.7.weight"].shape) == (1, 4)


def test_synthetic_weights_are_deterministic_and_follow_reference_init():
    model = build_model(model_config("meshed_memory_transformer", device="cpu"), SyntheticVocab())
    a = synthetic_state_dict(model.state_dict(), seed=1234)
    b = synthetic_state_dict(model.state_dict(), seed=1234)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert a["encoder.layers.0.mhatt.attention.fc_q.bias"].abs().max() == 0
    assert a["decoder.word_emb.components.weight"][0].abs().max() == 0
    bound = (6.0 / (512 + 512)) ** 0.5
    assert a["encoder.layers.0.mhatt.attention.fc_q.weight"].abs().max() <= bound
    assert abs(a["encoder.layers.0.mhatt.attention.m_k"].std().item() - 1 / 64) < 2e-3
    missing = model.load_state_dict(a, strict=False)
    assert not missing.unexpected_keys
    assert all(re.search(r"running_|pos_emb", k) for k in missing.missing_keys)


def test_instance_list_collates_ragged_rows_with_zero_padding():
    a = Instance(region_features=np.ones((3, 4), np.float32), region_boxes=torch.ones(3, 4), image_id=7)
    b = Instance(region_features=np.ones((5, 4), np.float32), region_boxes=torch.ones(5, 4), image_id=9)
    items = InstanceList([a, b])
    assert items.batch_size == 2 and tuple(items.region_features.shape) == (2, 5, 4)
    assert items.region_features[0, 3:].abs().sum() == 0 and items.region_features[0, :3].sum() == 12
    assert items.image_id == [7, 9] and items.missing_field is None
    assert items.to("cpu").region_boxes.shape == (2, 5, 4)


def test_statefulness_resets_even_on_error():
    model = build_model(model_config("standard_transformer", device="cpu"), SyntheticVocab())
    with pytest.raises(RuntimeError):
        with model.statefulness(4):
            assert model.decoder.running_seq.shape == (4, 1) and model.decoder._is_stateful
            assert model.decoder.layers[0].self_attn.running_keys.shape == (4, 0, 512)
            raise RuntimeError("boom")
    assert not model.decoder._is_stateful and model.decoder.running_seq.shape == (1,)
    assert model.encoder_features is None


def test_library_loads_and_exports_every_header_symbol():
    lib = native.load()
    header = open(os.path.join(REPO, "include", "ovc.h")).read()
    declared = set(re.findall(r"\b(ovc_[a-z_0-9]+)\s*\(", header))
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ovc_abi_version() == native.ABI_VERSION
    assert b"gfx950" in lib.ovc_build_info()


def test_product_path_fails_loudly_without_a_gpu_tensor():
    from openviic_amd import ops
    with pytest.raises(native.OvcError):
        ops.linear(torch.randn(4, 8), torch.randn(8, 8))
    model = build_model(model_config("standard_transformer", device="cpu"), SyntheticVocab())
    items = InstanceList()
    items.region_features = torch.randn(2, 50, 2048)
    with pytest.raises(native.OvcError):
        model.beam_search(items, batch_size=2, beam_size=5)


def test_product_never_imports_the_oracle():
    root = os.path.join(REPO, "openviic_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dirpath, f)
