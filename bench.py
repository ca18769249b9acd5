#!/usr/bin/env python3
"""Headline benchmark: captions/s of beam-5 decoding on synthetic region features.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config standard_transformer]
                    [--batch 256] [--beam 5] [--streams 4] [--tune-concurrency C] [--no-cpu-baseline]
                    [--precision f32] [--also-precision f16x3|none]

One "step" is one pass of the hot path over one batch: ``[B=256, 50, 2048]`` fp32 region features
already resident in HBM -> encoder -> 20 beam-search steps (beam 5, V=10201) -> token ids
``[B, 20]``, plus (N > 1) the one RCCL all-gather of the ids.  Images shard data-parallel: every
rank decodes its own B images (weak scaling), no data-path collective.  Rank 0 prints ONE JSON
line; ``value`` is whole-job captions/s (all ranks' images / max-over-ranks time).

The images go through ``openviic_amd.distributed.decode_sharded`` -- the code the world-size-2 gloo tests
cover -- at every N: each rank decodes rows ``shard_bounds(B*N, rank, N)`` of the global batch.

Timed region (``value``, ``ms_per_step``): consecutive batches alternate over ``--streams`` HIP streams, decode launch
sequences replayed as hipGraphs.  ``roofline`` covers the dominant kernel, the fp32 MFMA GEMM (every projection / FFN /
vocabulary product): algorithmic FLOPs 2*M*N*K per launch over the launch's own duration (kernel-scoped HIP events:
dispatch begin / end on the launch stream).  Durations of kernels that overlap on different streams do not add up to
wall time, so that leg runs on ONE stream, right after a timed single-stream region whose ``ms_per_step`` is reported
next to it (``roofline.single_stream``); ``roofline.timed_mode`` relates the same GEMM FLOPs to the headline wall time.
``cpu_baseline`` times the CPU oracle (which reproduces the reference's operation sequence): B = 256, one warm-up and
three timed repeats.

Everything above is fp32 MFMA arithmetic -- the parity mode and the only thing ``value`` ever means.  ``opt_in_precision``
(one GPU, after all fp32 legs) is a separately reported leg in an opt-in split-precision engine mode (default f16x3:
GEMMs on two fp16 planes of the fp32 operands, fp32 accumulation) with the fraction of the batch's
captions that come out identical to the fp32 engine's; ``--precision`` runs the whole bench in such a mode and says so in
``dtype``.
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
    ap.add_argument("--tune-concurrency", type=int, default=0,
                    help="GEMM tuning objective of the timed region's engine: tilings ranked by the time of this many "
                         "co-running copies (0 = the stream count, at most 4, with three or more streams, else 1).  Changes speed "
                         "only, never a bit.")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x6", "f16x3"],
                    help="GEMM arithmetic.  f32 (default) = fp32 MFMA, the parity mode and the only headline.  The others are the "
                         "opt-in split-precision modes (bf16 planes, fp32 accumulate), reported separately: NOT bit-identical to f32.")
    ap.add_argument("--also-precision", default="f16x3", choices=["none", "bf16x6", "f16x3"],
                    help="with --precision f32 on one GPU: an extra, separately reported leg in this opt-in mode "
                         "(same streams, same steps; never part of `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-measurement-hooks", action="store_true",
                    help="tools/ only: run although OVC_DEBUG_* / OVC_KSPLIT_* / a kernel A/B switch is set or the measurement build of "
                         "the library (tools/libovc_hooks.so) is loaded.  The line then says \"valid_for_credit\": false.")
    ap.add_argument("--cpu-sample", type=int, default=256, help="images per CPU-oracle repeat (BASELINE.md: B=256; ~12 s each on 16 cores)")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    return ap.parse_args()


PROFILE_VARIANT_TAGS = ("meshed_memory_transformer", "object_relation_transformer", "attention_on_attention")
PROFILE_PRECISION_TAGS = ("f16x3", "bf16x6", "bf16x3", "bf16")   # tags of committed profiles (round 2 had two more modes)
# compulsory HBM bytes per batch: features in + ids / log-probs out per caption, cross-K/V written
# once and read once per decoder layer and step is NOT compulsory (it could stay on chip), weights read once per batch
WEIGHT_BYTES = {"standard_transformer": 134.0e6, "standard_transformer_using_region": 134.0e6,
                "object_relation_transformer": 134.0e6}


def profiled_gemm_traffic(variant, precision="f32", profiles_dir=None):
    """HBM bytes moved by the GEMM launches of this workload, from the committed PMC passes (profiles/*_per_kernel_shape.csv,
    produced by tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled as the gfx950 guide
    prescribes).  Returns (bytes per GEMM launch, launch-weighted; file name; counter bytes per batch over ALL kernels of the
    file) or (None, None, None).  Not a live measurement.

    The file is chosen by workload AND arithmetic: ``r<NN><letter>[_<variant>][_<precision>]_per_kernel_shape.csv`` whose
    variant tag equals the benchmarked architecture (none = standard transformer) and whose precision tag equals the mode
    (none = fp32), and whose GEMM rows carry the kernel family of that mode; the newest such file wins."""
    import csv
    import glob
    import re
    family = "gemm_f32_mfma" if precision == "f32" else "gemm_split_mfma"
    want_variant = "" if variant.startswith("standard") else variant
    want_precision = "" if precision == "f32" else precision
    chosen = None
    for path in sorted(glob.glob(os.path.join(profiles_dir or os.path.join(REPO, "profiles"), "r[0-9][0-9]*_per_kernel_shape.csv"))):
        m = re.match(r"r\d\d[a-z]?_?(.*?)_?per_kernel_shape\.csv$", os.path.basename(path))
        if not m:
            continue
        tag = m.group(1)
        file_variant = next((v for v in PROFILE_VARIANT_TAGS if v in tag), "")
        rest = tag.replace(file_variant, "")
        file_precision = next((p for p in PROFILE_PRECISION_TAGS if p in rest), "")
        if file_variant != want_variant or file_precision != want_precision or "4streams" in rest:
            continue
        launches = total = everything = 0.0
        for row in csv.DictReader(open(path)):
            if not row["hbm_read_MB_per_launch"]:
                continue
            n = float(row["launches_per_batch"])
            moved = n * (float(row["hbm_read_MB_per_launch"]) + float(row["hbm_write_MB_per_launch"] or 0.0)) * 1048576.0
            everything += moved
            if row["kernel"].startswith(family):
                launches += n
                total += moved
        if launches:
            chosen = (round(total / launches, 0), os.path.basename(path), round(everything, 0))
    return chosen or (None, None, None)


def algorithmic_bytes(variant, B):
    """Compulsory HBM bytes of one batch: per caption 409 600 B of features in, 614 400 B of
    cross-attention K/V written once, 240 B of ids and log-probs out; the weights once per batch."""
    weights = WEIGHT_BYTES.get(variant)
    return None if weights is None else float(B * (409600 + 614400 + 240) + weights)


# Switches that change result bits, skip work or select another kernel.  The shipped library does not read them at all
# (csrc/common.h: they exist only in the -DOVC_MEASUREMENT_HOOKS build of tools/); a benchmark line produced with one of
# them set, or with that build loaded, is not a measurement of the product.
HOOK_PREFIXES = ("OVC_DEBUG_", "OVC_KSPLIT_")
HOOK_SWITCHES = ("OVC_SELECT_TWO_PASS", "OVC_VOCAB_ROW_MAJOR", "OVC_K1_SEPARATE", "OVC_SELF_ATTENTION_ROWS", "OVC_ATTENTION_GENERAL")
RECORDED_PREFIXES = ("OVC_", "GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "HSA_", "ROCR_", "HIP_VISIBLE", "CUDA_VISIBLE", "NCCL_", "RCCL_")


def measurement_hooks_in(environ, build_info=""):
    """Names of the measurement hooks active for this process: environment switches and the hooks build itself."""
    found = sorted(k for k in environ if k.startswith(HOOK_PREFIXES) or k in HOOK_SWITCHES)
    if "measurement-hooks" in build_info:
        found.append("library built with -DOVC_MEASUREMENT_HOOKS")
    return found


def recorded_environment(environ):
    """What the line says it ran with: every OVC_* / HIP-runtime variable that can influence speed or the engine's mode."""
    return {k: environ[k] for k in sorted(environ) if k.startswith(RECORDED_PREFIXES)}


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


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(cfg, sd, variant, beam, sample, repeats):
    """Oracle (kind "port": reference op sequence restated on PyTorch-CPU) on the host cores, BASELINE.md section 3:
    same weights and inputs as the GPU run, B = ``sample`` images per call, 1 warm-up + ``repeats`` timed calls, median."""
    from oracle.captioner import OracleCaptioner        # checker / baseline only, never the product path
    cores = usable_cores()
    torch.set_num_threads(cores)
    cpu = cpu_model_string()
    print("[bench] cpu baseline: %d images x (1 + %d) calls on %d threads of %s (host reports %d cpus)"
          % (sample, repeats, cores, cpu, os.cpu_count() or 0), file=sys.stderr, flush=True)
    oracle = OracleCaptioner(cfg, sd, V, T)
    feats = synthetic_features(sample, N_REGIONS, D_FEAT, seed=0)
    boxes = synthetic_boxes(sample, N_REGIONS, seed=0) if variant == "object_relation_transformer" else None
    times = []
    for rep in range(repeats + 1):
        t0 = time.perf_counter()
        oracle.beam_search(feats, beam, boxes=boxes)
        dt = time.perf_counter() - t0
        print("[bench] cpu baseline %s: %.2f s" % ("warm-up" if rep == 0 else "repeat %d" % rep, dt), file=sys.stderr, flush=True)
        if rep:
            times.append(dt)
    # BASELINE config 1 (the reference's own CPU-runnable case): greedy decode, B = 4, same weights, 1 warm-up + 5 timed calls
    greedy = []
    for rep in range(6):
        t0 = time.perf_counter()
        oracle.beam_search(feats[:4], 1, boxes=None if boxes is None else boxes[:4])
        if rep:
            greedy.append(time.perf_counter() - t0)
    return {"value": round(sample / statistics.median(times), 3), "unit": "captions/s", "cores": cores,
            "kind": "port", "cpu": cpu,
            "config1_greedy_b4": {"value": round(4 / statistics.median(greedy), 2), "unit": "captions/s",
                                  "sample": "B=4, beam 1 (greedy), 1 warm-up + 5 timed calls, median %.3f s" % statistics.median(greedy)},
            "sample": "B=%d images per call, beam %d, same weights/inputs as the GPU run, 1 warm-up + %d timed calls "
                      "(median; min %.2f s, max %.2f s), torch %s CPU fp32, %d threads"
                      % (sample, beam, repeats, min(times), max(times), torch.__version__, cores)}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: libraries that chat on stdout (RCCL prints a version banner when its first
    # communicator comes up) are sent to stderr until the line is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or "RANK" in os.environ      # under torch.distributed.run the RCCL path runs even for N = 1
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the engine has no CPU path")
    from openviic_amd import native as _native
    build_info = _native.load().ovc_build_info().decode()
    hooks = measurement_hooks_in(os.environ, build_info)
    if hooks and not args.allow_measurement_hooks:
        raise SystemExit("bench.py: measurement hooks are active (%s): results would not be the product's.  Unset them / load the "
                         "default library, or pass --allow-measurement-hooks (tools/ only; the line is then marked invalid for credit)."
                         % ", ".join(hooks))
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

    # every rank holds the same global batch in HBM; decode_sharded hands each rank its contiguous shard
    from openviic_amd.distributed import decode_sharded
    feats = synthetic_features(B * world, N_REGIONS, D_FEAT, seed=0).to(device)
    boxes = synthetic_boxes(B * world, N_REGIONS, seed=0).to(device) if variant == "object_relation_transformer" else None

    def decode(f, b):
        items = InstanceList()
        items.region_features = f
        if b is not None:
            items.region_boxes = b
        return model.beam_search(items, batch_size=f.shape[0], beam_size=k, out_size=1)

    streams = [torch.cuda.Stream(device=device) for _ in range(max(1, args.streams))]
    issued = [0]
    # Tilings are ranked for the mode they run in: with several batches in flight a GEMM shares the chip with other
    # streams' kernels, so the timed region's engine consults the table measured with two co-running copies (fewer,
    # larger tiles: +2.5..3 % captions/s, same box, alternating runs); the single-stream leg below -- the mode the
    # kernel-scoped roofline belongs to -- uses the table measured in isolation.  All tilings of a K-order class give the
    # same bits, so this moves no result.
    from openviic_amd.engine import CaptionEngine
    # (round 3, one box, two alternating rounds: objective 1 / 2 / 3 / 4 = 22.8-23.0k / 23.2-23.3k / 23.3-23.4k / 23.44-23.46k)
    objective = args.tune_concurrency or (min(len(streams), 4) if len(streams) >= 3 else 1)
    engine_timed = CaptionEngine(model, tune_concurrency=objective, precision=args.precision)
    engine_single = engine_timed if objective == 1 else CaptionEngine(model, tune_concurrency=1, precision=args.precision)
    model._engine = engine_timed

    def step(slot=None):
        # consecutive batches are independent: alternate them over the streams (each stream has its own
        # engine workspace), so a batch's small decode launches overlap the other batch's.  The path's one exchange --
        # the all-gather of every rank's token ids (RCCL over xGMI, 40 KB per rank, once per batch, on the decoding
        # stream) -- happens inside decode_sharded.
        slot = issued[0] % len(streams) if slot is None else slot
        issued[0] += 1
        with torch.cuda.stream(streams[slot]):
            return decode_sharded(decode, feats, boxes)

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

    # ---- single-stream leg: a timed region on ONE stream, then the instrumented pass on the same stream -------------
    single_steps = max(3, args.steps // 3)
    with torch.no_grad():
        if engine_single is not engine_timed:
            model._engine = engine_single
            for _ in range(3):                      # its own set-up: tiling measurement, plain pass, graph capture
                step(0)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(single_steps):
            step(0)
        torch.cuda.synchronize()
        single_ms = 1e3 * (time.perf_counter() - t0) / single_steps

    result = None
    if rank == 0:
        from openviic_amd import native
        import ctypes
        lib = native.load()
        # kernel-scoped events around every GEMM launch of one batch, on the launch stream (plain launches: a replayed
        # graph cannot carry per-kernel events)
        lib.ovc_profile_enable(1)
        with torch.no_grad(), torch.cuda.stream(streams[0]):
            decode(feats[:B], None if boxes is None else boxes[:B])
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
        traffic, traffic_source, counter_bytes = (profiled_gemm_traffic(variant, args.precision) if B == 256 and k == 5
                                                  else (None, None, None))
        split_products = {"f32": 0, "bf16x6": 6, "f16x3": 3}[args.precision]
        if split_products:      # opt-in mode: the peak is the bf16 dense MFMA rate shared by the plane products of one fp32 product
            PEAK = round(PEAK_F32_MFMA_TFLOPS * 16 / split_products, 1)
            kernel_label = ("gemm_split_mfma<BM,BN,WM,WN,BK,MODE> (v_mfma_f32_32x32x16_bf16 / _f16, %d plane products per product; "
                            "achieved / peak in fp32-product-equivalent TFLOP/s), all tilings" % split_products)
        else:
            PEAK = PEAK_F32_MFMA_TFLOPS
            kernel_label = "gemm_f32_mfma<BM,BN,WM,WN,WK,BK> (v_mfma_f32_32x32x2_f32), all tilings"
        roofline = {"bound": "mfma", "kernel": kernel_label,
                    "achieved": round(all_gemm, 2), "peak": PEAK, "unit": "TFLOP/s",
                    "frac": round(all_gemm / PEAK, 4), "traffic": traffic,
                    "traffic_source": ("HBM bytes per launch (read + write), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: profiles/%s"
                                       % traffic_source) if traffic else None,
                    # waste indicator: counter bytes of ALL kernels of one batch over the compulsory bytes of one batch
                    # (features in, cross-K/V written once, ids / log-probs out, weights once: SURVEY.md section 8d)
                    "traffic_per_batch": counter_bytes,
                    "algorithmic_bytes_per_batch": algorithmic_bytes(variant, B),
                    "traffic_ratio": (round(counter_bytes / algorithmic_bytes(variant, B), 1)
                                      if counter_bytes and algorithmic_bytes(variant, B) else None),
                    "launches_per_step": tot_n, "avg_launch_us": round(1e3 * tot_ms / max(tot_n, 1), 2),
                    "flops_per_launch": round(tot_fl / max(tot_n, 1), 0), "kernel_ms_per_step": round(tot_ms, 3),
                    "timing": "hipExtLaunchKernelGGL start/stop events (dispatch begin/end timestamps) on the launch "
                              "stream for every GEMM launch of one batch decoded on a single stream",
                    # the mode the kernel durations belong to: one stream, its own timed region of `steps` batches
                    "single_stream": {"steps": single_steps, "ms_per_step": round(single_ms, 3),
                                      "captions_per_s": round(B * world / single_ms * 1e3, 1),
                                      "gemm_share_of_step": round(tot_ms / single_ms, 4)},
                    # the headline mode: the same GEMM FLOPs per batch over the timed region's wall time per batch
                    "timed_mode": {"streams": len(streams), "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                                   "gemm_tflops_over_wall": round(tot_fl / (1e3 * elapsed / args.steps) / 1e9, 2),
                                   "frac": round(tot_fl / (1e3 * elapsed / args.steps) / 1e9 / PEAK, 4)},
                    "per_kernel": per_kernel, "per_class": per_class}
        # (under a profiler the instrumented pass is slowed down more than the graph replay: flagged, not fatal)
        roofline["single_stream"]["kernel_time_within_step"] = bool(tot_ms <= single_ms * 1.02)
        if tot_ms > single_ms * 1.02:
            print("[bench] WARNING: GEMM kernel time %.3f ms exceeds the single-stream step %.3f ms (profiler attached?)"
                  % (tot_ms, single_ms), file=sys.stderr, flush=True)
        # K1: the path's one pass over the caller's features.  Since round 3 the padding mask is found by the
        # feature projection itself while it stages its A tiles (GemmArgs::zero_rows_out), so the pass IS that GEMM launch: its
        # algorithmic bytes (features in, weight in, projected rows + mask out) over its kernel-scoped duration from the same
        # instrumented batch as above.  (Rounds 1-3 timed the stand-alone mask kernel here, 23 times over a tensor that fits the
        # Infinity Cache -- a kernel the fp32 engine no longer launches.)  The launch is matrix-bound, not HBM-bound: both shown.
        fp = per_class.get("feature_proj")
        if fp:
            d_model = int(cfg.VISION_EMBEDDING.D_MODEL)
            fp_bytes = 4.0 * (B * N_REGIONS * D_FEAT + d_model * D_FEAT + B * N_REGIONS * d_model) + B * N_REGIONS
            fp_us = 1e3 * fp["ms"] / fp["launches"]
            roofline["k1_feature_pass"] = {
                "kernel": "the feature projection GEMM (gemm_f32_mfma, M = B*N, K = d_feat) incl. the padding mask it finds while staging A",
                "bound": "mfma", "bytes_per_launch": fp_bytes, "avg_us": round(fp_us, 2),
                "achieved": fp["tflops"], "peak": PEAK, "unit": "TFLOP/s", "frac": round(fp["tflops"] / PEAK, 4),
                "hbm": {"achieved": round(fp_bytes / fp_us / 1e3, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(fp_bytes / fp_us / 1e3 / 8000.0, 4)}}
        if gflop:
            e2e = captions_per_s / world * gflop / 1e3
            roofline["end_to_end"] = {"gflop_per_caption": gflop, "achieved": round(e2e, 2),
                                      "frac": round(e2e / PEAK, 4)}
        # ---- opt-in split-precision leg (reported separately, never `value`): the same timed region on an engine whose GEMMs
        # run on 16-bit planes, plus the fraction of this batch's captions that come out token-for-token as in fp32 ------------
        also = None
        if world == 1 and args.precision == "f32" and args.also_precision != "none":
            try:
                with torch.no_grad():
                    model._engine = engine_timed
                    ids_f32 = step(0)
                    ids_f32 = ids_f32[0] if isinstance(ids_f32, tuple) else ids_f32
                    model._engine = CaptionEngine(model, tune_concurrency=objective, precision=args.also_precision)
                    for _ in range(3 * len(streams)):
                        step()
                    torch.cuda.synchronize()
                    for _ in range(args.warmup):
                        step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
                    also_elapsed = time.perf_counter() - t0
                    ids_alt = step(0)
                    ids_alt = ids_alt[0] if isinstance(ids_alt, tuple) else ids_alt
                    torch.cuda.synchronize()
                    model._engine = engine_timed
                also = {"mode": args.also_precision, "value": round(B * args.steps / also_elapsed, 2), "unit": "captions/s",
                        "ms_per_step": round(1e3 * also_elapsed / args.steps, 3), "steps": args.steps, "streams": len(streams),
                        "captions_identical_to_f32": round(float((ids_alt == ids_f32).all(dim=1).float().mean().item()), 4),
                        "note": "opt-in engine mode (CaptionEngine(precision=...) / OVC_PRECISION): every GEMM contracts 16-bit planes of "
                                "its fp32 operands with fp32 accumulation; fp32 in/out; not bit-identical to the fp32 path, not the headline"}
                print("[bench] opt-in %s: %.1f captions/s, %.2f ms/step, %.1f %% of this batch's captions identical to fp32's"
                      % (also["mode"], also["value"], also["ms_per_step"], 100 * also["captions_identical_to_f32"]), file=sys.stderr, flush=True)

            except Exception as exc:          # the separately reported leg must never cost the headline its JSON line
                model._engine = engine_timed
                also = {"mode": args.also_precision, "error": "%s: %s" % (type(exc).__name__, exc)}
                print("[bench] opt-in %s leg failed: %s" % (args.also_precision, also["error"]), file=sys.stderr, flush=True)
        result = {
            "metric": "captions/sec (whole node) at beam=%d, %d regions x d%d" % (k, N_REGIONS, D_FEAT),
            "value": round(captions_per_s, 2), "unit": "captions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not split_products else "f32 in/out, GEMMs as %d 16-bit plane products (%s): opt-in mode, not the parity mode" % (split_products, args.precision),
            "data": "synthetic",
            "config": {"workload": "%s beam=%d, B=%d per GPU, %dx%d synthetic regions, V=%d, max_len=%d, "
                                   "random-init weights" % (variant, k, B, N_REGIONS, D_FEAT, V, T),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "streams": len(streams),
                       "gemm_tuning_objective": {"timed_region": objective, "single_stream_leg": 1},
                       "library": build_info, "environment": recorded_environment(os.environ)},
            "roofline": roofline,
        }
        if hooks:
            result["measurement_hooks"] = hooks
            result["valid_for_credit"] = False
        if also:
            result["opt_in_precision"] = also
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, sd, variant, k, args.cpu_sample, args.cpu_repeats)
        elif world > 1:
            # the bench contract times the CPU path on rank 0 at N = 1 only; the N > 1 lines point there instead of dropping the key
            result["cpu_baseline"] = {"value": None, "unit": "captions/s", "kind": "port",
                                      "sample": "not timed at N > 1: see the N = 1 line of the same run (bench contract: rank 0, N = 1 only)"}
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(result), flush=True)
        os.dup2(2, 1)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
