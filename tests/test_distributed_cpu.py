"""The N > 1 path (shard -> decode -> all-gather of token ids) over gloo, world_size 2 and 4, on CPU.

The decode function is the CPU oracle here (the HIP engine needs a GPU); what is under test is the
sharding, padding and the collective in ``openviic_amd.distributed``.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import tiny_case
from oracle.captioner import OracleCaptioner


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openviic_amd.distributed import decode_sharded
    cfg, vocab, sd, feats, _ = tiny_case("standard_transformer", B=total)
    oracle = OracleCaptioner(cfg, sd, len(vocab), vocab.max_caption_length)
    calls = []

    def decode(f, b):
        calls.append(f.shape[0])
        return oracle.beam_search(f, 3, boxes=b)
    ids, logp = decode_sharded(decode, feats, gather_log_probs=True)
    np.save(os.path.join(out_dir, "ids_%d.npy" % rank), ids.numpy())
    np.save(os.path.join(out_dir, "logp_%d.npy" % rank), logp.numpy())
    np.save(os.path.join(out_dir, "calls_%d.npy" % rank), np.array(calls))
    dist.barrier()
    dist.destroy_process_group()


def _run(total, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    cfg, vocab, sd, feats, _ = tiny_case("standard_transformer", B=total)
    want_ids, want_logp = OracleCaptioner(cfg, sd, len(vocab), vocab.max_caption_length).beam_search(feats, 3)
    for rank in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("ids_%d.npy" % rank)), want_ids.numpy())
        np.testing.assert_allclose(np.load(tmp_path / ("logp_%d.npy" % rank)), want_logp.numpy(), rtol=1e-5, atol=1e-6)
        assert np.load(tmp_path / ("calls_%d.npy" % rank)).tolist() == [(total + world - 1) // world]


def test_even_shards(tmp_path):
    _run(6, tmp_path)


def test_ragged_last_shard_is_padded(tmp_path):
    _run(5, tmp_path)


def test_four_ranks_with_a_short_and_an_empty_shard(tmp_path):
    """BASELINE config 5 shards over 8 GPUs; more than two ranks exercise the ordering of the gathered shards, and 5 images over
    4 ranks leave rank 2 one image and rank 3 NONE: its whole shard is zero-row padding images, decoded and dropped."""
    _run(5, tmp_path, world=4)


def test_four_ranks_even(tmp_path):
    _run(8, tmp_path, world=4)


def test_shard_bounds():
    from openviic_amd.distributed import shard_bounds
    assert shard_bounds(2048, 3, 8) == (768, 1024, 256)
    assert shard_bounds(5, 1, 2) == (3, 5, 3)
    assert shard_bounds(3, 3, 4) == (3, 3, 1)
    covered = sorted(i for r in range(8) for i in range(*shard_bounds(1001, r, 8)[:2]))
    assert covered == list(range(1001))


def test_single_process_passthrough():
    from openviic_amd.distributed import decode_sharded
    out = decode_sharded(lambda f, b: (torch.arange(f.shape[0])[:, None], torch.zeros(f.shape[0], 1)), torch.zeros(4, 2, 3))
    assert out.squeeze(1).tolist() == [0, 1, 2, 3]
