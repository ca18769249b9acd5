"""The N > 1 path with the HIP engine on every rank: ranks of one box share its GPU (the GPU box has one), each decodes its shard
on the engine and the token ids are all-gathered.  RCCL refuses two ranks on one device, so the collective runs over gloo on host
copies -- what is under test is the engine inside ``decode_sharded`` in several processes at once: shard bounds, zero-row padding
images on the last rank, and that a shard decodes exactly as it does inside the whole batch (``bench.py --gpus N`` relies on it).
"""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import batch, device_model, full_case

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openviic_amd.distributed import decode_sharded
    cfg, vocab, sd, feats, _ = full_case("standard_transformer", total, ragged=True)
    model = device_model(cfg, vocab, sd)
    calls = []

    def decode(f, b):
        calls.append(f.shape[0])
        with torch.no_grad():
            ids, logp = model.beam_search(batch(f, b), batch_size=f.shape[0], beam_size=5)
        return ids.cpu(), logp.cpu()
    ids, logp = decode_sharded(decode, feats, gather_log_probs=True)
    np.save(os.path.join(out_dir, "ids_%d.npy" % rank), ids.numpy())
    np.save(os.path.join(out_dir, "logp_%d.npy" % rank), logp.numpy())
    np.save(os.path.join(out_dir, "calls_%d.npy" % rank), np.array(calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 24), (3, 20)])
def test_ranks_sharing_the_gpu_decode_their_shards_like_the_whole_batch(tmp_path, world, total):
    ctx = mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=False)
    deadline = time.time() + 300                       # a rank that never reaches the rendezvous must not hang the suite
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.terminate()
            pytest.fail("ranks did not finish within 300 s")
    cfg, vocab, sd, feats, _ = full_case("standard_transformer", total, ragged=True)
    model = device_model(cfg, vocab, sd)
    with torch.no_grad():
        want_ids, want_logp = model.beam_search(batch(feats), batch_size=total, beam_size=5)
    for rank in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("ids_%d.npy" % rank)), want_ids.cpu().numpy())
        np.testing.assert_array_equal(np.load(tmp_path / ("logp_%d.npy" % rank)), want_logp.cpu().numpy())   # bit-identical: fixed K order
        assert np.load(tmp_path / ("calls_%d.npy" % rank)).tolist() == [(total + world - 1) // world]
