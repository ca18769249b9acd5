#!/usr/bin/env python3
"""Headline benchmark: captions/s of beam-5 decoding on synthetic region features.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config standard_transformer]
                    [--batch 256] [--beam 5] [--no-cpu-baseline]

One "step" is one pass of the hot path over one batch: ``[B=256, 50, 2048]`` fp32 region features
already resident in HBM -> encoder -> 20 beam-search steps (beam 5, V=10201) -> token ids
``[B, 20]``, plus (N > 1) the one RCCL all-gather of the ids.  Images shard data-parallel: every
rank decodes its own B images (weak scaling), no data-path collective.  Rank 0 prints ONE JSON
line; ``value`` is whole-job captions/s (all ranks' images / max-over-ranks time).

``roofline`` covers the dominant kernel, the fp32 MFMA GEMM (every projection / FFN / vocabulary
product): algorithmic FLOPs 2*M*N*K per launch over its hipEvent-bracketed duration on the launch
stream, measured in an extra instrumented pass after the timed region.  ``cpu_baseline`` times the
CPU oracle (which reproduces the reference's operation sequence) on a bounded sample.
"""
import argparse
import json
import os
import statistics
import sys
import time

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): the four decode streams plus
# torch's default stream need more, or two of them share a queue and serialise (tools/queue_probe.sh: 4 streams
# give 0.93x of 3 streams on 4 queues and 1.025x on 8).  Read when the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch                                                                       # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from openviic_amd.builders import build_model                                   # noqa: E402
from openviic_amd.config import model_config                                    # noqa: E402
from openviic_amd.instance import InstanceList                                  # noqa: E402
from openviic_amd.utils.synthetic import (SyntheticVocab, synthetic_boxes, synthetic_features,   # noqa: E402
                                          synthetic_state_dict)

V, T, N_REGIONS, D_FEAT = 10201, 20, 50, 2048
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
GFLOP_PER_CAPTION = {"standard_transformer": 4.374, "standard_transformer_using_region": 4.374,
                     "attention_on_attention": None, "object_relation_transformer": 4.374,
                     "meshed_memory_transformer": 6.270}     # SURVEY.md section 8d (minimal algorithm)
GEMM_CLASSES = ["feature_proj", "encoder", "decoder_proj_ffn", "vocab"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", default="standard_transformer")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams that consecutive (independent) batches alternate on; decode steps are "
                         "small launches, so several batches in flight fill the chip better than one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=128, help="images in the CPU-oracle sample (about 13 s on 16 cores)")
    return ap.parse_args()


def profiled_gemm_traffic():
    """HBM bytes per GEMM launch from the committed PMC passes (profiles/*_per_kernel_shape.csv, produced by
    tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as the gfx950 guide
    prescribes), launch-weighted over the GEMM rows.  Not a live measurement: None when the file is absent."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_per_kernel_shape.csv")))
    if not files:
        return None, None
    launches = total = 0.0
    for row in csv.DictReader(open(files[-1])):
        if not row["kernel"].startswith("gemm_f32_mfma") or not row["hbm_read_MB_per_launch"]:
            continue
        n = float(row["launches_per_batch"])
        launches += n
        total += n * (float(row["hbm_read_MB_per_launch"]) + float(row["hbm_write_MB_per_launch"] or 0.0)) * 1048576.0
    if not launches:
        return None, None
    return round(total / launches, 0), os.path.basename(files[-1])


def usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, not the host total
    (a GPU box hands a 1-GPU job a share of the host, and oversubscribing it stalls OpenMP)."""
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    cores = min(cores, max(1, int(int(fields[0]) / int(fields[1]))))
            else:
                quota = int(fields[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    cores = min(cores, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(cores, 32))


def cpu_baseline(cfg, sd, variant, beam, sample):
    """Oracle (kind "port": reference op sequence restated on PyTorch-CPU) on the host cores."""
    from oracle.captioner import OracleCaptioner        # checker / baseline only, never the product path
    cores = usable_cores()
    torch.set_num_threads(cores)
    print("[bench] cpu baseline: %d images on %d threads (host reports %d cpus)" % (sample, cores, os.cpu_count() or 0),
          file=sys.stderr, flush=True)
    oracle = OracleCaptioner(cfg, sd, V, T)
    feats = synthetic_features(sample, N_REGIONS, D_FEAT, seed=0)
    boxes = synthetic_boxes(sample, N_REGIONS, seed=0) if variant == "object_relation_transformer" else None
    oracle.beam_search(feats[:4], beam, boxes=None if boxes is None else boxes[:4])       # warm-up
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        oracle.beam_search(feats, beam, boxes=boxes)
        times.append(time.perf_counter() - t0)
        print("[bench] cpu baseline repeat: %.2f s" % times[-1], file=sys.stderr, flush=True)
    return {"value": round(sample / statistics.median(times), 3), "unit": "captions/s", "cores": cores,
            "kind": "port",
            "sample": "%d images, beam %d, same weights/inputs, 1 warm-up + 2 timed repeats (median), "
                      "torch %s CPU fp32, %d threads" % (sample, beam, torch.__version__, cores)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or "RANK" in os.environ      # under torch.distributed.run the RCCL path runs even for N = 1
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the engine has no CPU path")
    # one rank per GPU; if the launcher narrows each rank's visible devices to its own GPU, LOCAL_RANK still counts up
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    variant, B, k = args.config, args.batch, args.beam
    vocab = SyntheticVocab(V, T)
    cfg = model_config(variant, d_feature=D_FEAT, device=str(device))
    model = build_model(cfg, vocab).eval()
    sd = synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init")
    model.load_state_dict(sd, strict=False)

    # every rank draws the same global batch and takes its contiguous shard
    feats = synthetic_features(B * world, N_REGIONS, D_FEAT, seed=0)[rank * B:(rank + 1) * B]
    items = InstanceList()
    items.region_features = feats.to(device)
    if variant == "object_relation_transformer":
        items.region_boxes = synthetic_boxes(B * world, N_REGIONS, seed=0)[rank * B:(rank + 1) * B].to(device)


    streams = [torch.cuda.Stream(device=device) for _ in range(max(1, args.streams))]
    # one gather buffer per stream: batches in flight on different streams never share an output
    gathered = [torch.empty(world * B, T, dtype=torch.int64, device=device) for _ in streams] if distributed else None
    issued = [0]

    def step():
        # consecutive batches are independent: alternate them over the streams (each stream has its own
        # engine workspace), so a batch's small decode launches overlap the other batch's
        slot = issued[0] % len(streams)
        stream = streams[slot]
        issued[0] += 1
        with torch.cuda.stream(stream):
            ids, _ = model.beam_search(items, batch_size=B, beam_size=k, out_size=1)
            if distributed:
                # the path's one exchange: token ids of every rank for evaluation (RCCL all-gather over xGMI,
                # 40 KB per rank, on the decoding stream, once per batch)
                dist.all_gather_into_tensor(gathered[slot], ids.contiguous())
        return ids

    with torch.no_grad():
        # engine set-up, not measurement: the first call of a (shape, stream) tunes the GEMM tilings and warms every
        # kernel, the second captures the decode launch sequence as a hipGraph; from the third on it is replayed
        for _ in range(3 * len(streams)):
            step()
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        from openviic_amd import native
        import ctypes
        lib = native.load()
        # ---- instrumented pass: hipEvents around every GEMM launch, on the launch stream ----------
        lib.ovc_profile_kernel_name.restype = ctypes.c_char_p
        lib.ovc_profile_enable(1)
        with torch.no_grad():
            model.beam_search(items, batch_size=B, beam_size=k, out_size=1)
        torch.cuda.synchronize()
        lib.ovc_profile_enable(0)
        def read(kind, index):
            n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            lib.ovc_profile_read(kind, index, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
            return n.value, ms.value, fl.value

        def entry(n, ms, fl):
            return {"launches": n, "ms": round(ms, 4), "avg_us": round(1e3 * ms / n, 2), "tflops": round(fl / ms / 1e9, 2)}
        per_class, per_kernel, tot_n, tot_ms, tot_fl = {}, {}, 0, 0.0, 0.0
        for c, name in enumerate(GEMM_CLASSES):
            n, ms, fl = read(0, c)
            if n:
                per_class[name] = entry(n, ms, fl)
            tot_n += n; tot_ms += ms; tot_fl += fl
        t = 0
        while lib.ovc_profile_kernel_name(t):
            n, ms, fl = read(1, t)
            if n:
                per_kernel[lib.ovc_profile_kernel_name(t).decode()] = entry(n, ms, fl)
            t += 1
        captions_per_s = B * world * args.steps / elapsed
        # The dominant kernel is the fp32 MFMA GEMM template (every projection / FFN / vocabulary product):
        # >= 97 % of the algorithmic FLOPs and ~3/4 of the device time.  Its tilings are specialisations of one
        # kernel, so the roofline is taken over all of its launches in one batch; per-instance rows (names as
        # rocprofv3 prints them) are kept for cross-checking against profiles/*_kernel_stats.csv.
        all_gemm = tot_fl / tot_ms / 1e9 if tot_ms else 0.0
        print("[bench] gpu: %.1f captions/s, %.2f ms/step; GEMM %.2f TFLOP/s over %d launches (%.1f us avg, kernel-scoped events)"
              % (captions_per_s, 1e3 * elapsed / args.steps, all_gemm, tot_n, 1e3 * tot_ms / max(tot_n, 1)),
              file=sys.stderr, flush=True)
        gflop = GFLOP_PER_CAPTION.get(variant)
        traffic, traffic_source = profiled_gemm_traffic() if variant.startswith("standard") and B == 256 and k == 5 else (None, None)
        roofline = {"bound": "mfma", "kernel": "gemm_f32_mfma<BM,BN,WM,WN,WK,BK> (v_mfma_f32_32x32x2_f32), all tilings",
                    "achieved": round(all_gemm, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(all_gemm / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": ("HBM bytes per launch (read + write), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: profiles/%s"
                                       % traffic_source) if traffic else None,
                    "launches_per_step": tot_n, "avg_launch_us": round(1e3 * tot_ms / max(tot_n, 1), 2),
                    "flops_per_launch": round(tot_fl / max(tot_n, 1), 0), "kernel_ms_per_step": round(tot_ms, 3),
                    "timing": "hipExtLaunchKernelGGL start/stop events (dispatch begin/end timestamps) on the launch "
                              "stream for every GEMM launch of one instrumented batch after the timed region",
                    "per_kernel": per_kernel, "per_class": per_class}
        # K1: the padding-mask kernel is the path's one HBM-bound pass over the
        # features (B*N*d_feat fp32 in, B*N bytes out); torch events on the stream it is launched on.
        from openviic_amd import ops
        feats = items.region_features
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.zero_row_mask(feats)
        ev0.record()
        for _ in range(20):
            ops.zero_row_mask(feats)
        ev1.record()
        torch.cuda.synchronize()
        k1_us = ev0.elapsed_time(ev1) / 20 * 1e3
        k1_bytes = feats.numel() * 4 + feats.shape[0] * feats.shape[1]
        roofline["k1_hbm"] = {"kernel": "zero_row_mask_kernel", "bound": "hbm", "bytes_per_launch": k1_bytes,
                              "avg_us": round(k1_us, 2), "achieved": round(k1_bytes / k1_us / 1e3, 1), "peak": 8000.0,
                              "unit": "GB/s", "frac": round(k1_bytes / k1_us / 1e3 / 8000.0, 4)}
        if gflop:
            e2e = captions_per_s / world * gflop / 1e3
            roofline["end_to_end"] = {"gflop_per_caption": gflop, "achieved": round(e2e, 2),
                                      "frac": round(e2e / PEAK_F32_MFMA_TFLOPS, 4)}
        result = {
            "metric": "captions/sec (whole node) at beam=%d, %d regions x d%d" % (k, N_REGIONS, D_FEAT),
            "value": round(captions_per_s, 2), "unit": "captions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s beam=%d, B=%d per GPU, %dx%d synthetic regions, V=%d, max_len=%d, "
                                   "random-init weights" % (variant, k, B, N_REGIONS, D_FEAT, V, T),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "streams": len(streams)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, sd, variant, k, args.cpu_sample)
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
