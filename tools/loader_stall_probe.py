#!/usr/bin/env python3
"""What on the HOST slows the decode of a resident batch down?  (GPU box)

The loader-fed prediction loop decoded a B = 256 batch in 180-300 ms instead of 11-14: this probe decodes the same resident batch
on two streams in a loop (GPU events around every decode) while the host does one thing at a time:

  idle                    nothing else
  burners N               N processes spinning in Python (no memory traffic)
  memcpy thread           a thread of THIS process copying 105 MB into pinned memory in a loop (numpy, GIL released)
  memcpy process          the same copy loop in a child process (pageable -> pageable)
  h2d thread              a thread issuing 105 MB pinned -> device copies on its own stream
  gil thread              a thread spinning in pure Python (holds the GIL between the launching thread's calls)
  fork                    a thread calling fork() (child exits at once) every 50 ms; subprocess (vfork + exec) likewise

    python tools/loader_stall_probe.py
"""
import multiprocessing as mp
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def burn(stop):
    x = 0
    while not stop.is_set():
        for _ in range(100000):
            x += 1


def memcpy_process(stop):
    a = np.ones(256 * 50 * 2048, dtype=np.float32)
    b = np.empty_like(a)
    while not stop.is_set():
        np.copyto(b, a)


def main():
    from openviic_amd.builders import build_model
    from openviic_amd.config import model_config
    from openviic_amd.instance import InstanceList
    from openviic_amd.utils.synthetic import SyntheticVocab, synthetic_features, synthetic_state_dict
    V, T, B = 10201, 20, 256
    vocab = SyntheticVocab(V, T)
    model = build_model(model_config("standard_transformer", d_feature=2048, device="cuda:0"), vocab).eval()
    model.load_state_dict(synthetic_state_dict(model.state_dict(), seed=1234, mode="reference_init"), strict=False)
    items = InstanceList()
    items.region_features = synthetic_features(B, 50, 2048, seed=0).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def decode_loop(n=24):
        times = []
        with torch.no_grad():
            evs = []
            t0 = time.perf_counter()
            for i in range(n):
                s = streams[i % 2]
                with torch.cuda.stream(s):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(s)
                    model.beam_search(items, batch_size=B, beam_size=5)
                    b.record(s)
                    evs.append((a, b))
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        for a, b in evs[4:]:
            times.append(a.elapsed_time(b))
        times.sort()
        return times[len(times) // 2], wall / n * 1e3

    decode_loop(8)
    print("%-28s  median GPU ms per batch (event to event) / wall ms per batch" % "host activity")
    print("%-28s  %7.2f / %7.2f" % (("idle",) + decode_loop()), flush=True)

    ctx = mp.get_context("spawn")
    for n in (4, 8, 12):
        stop = ctx.Event()
        procs = [ctx.Process(target=burn, args=(stop,), daemon=True) for _ in range(n)]
        for p in procs:
            p.start()
        time.sleep(3.0)
        print("%-28s  %7.2f / %7.2f" % (("burners %d" % n,) + decode_loop()), flush=True)
        stop.set()
        for p in procs:
            p.join()

    stop_t = threading.Event()

    def run_thread(target, name):
        stop_t.clear()
        th = threading.Thread(target=target, daemon=True)
        th.start()
        time.sleep(0.5)
        print("%-28s  %7.2f / %7.2f" % ((name,) + decode_loop()), flush=True)
        stop_t.set()
        th.join()

    src = np.ones(B * 50 * 2048, dtype=np.float32)
    pinned = torch.empty(B * 50 * 2048, dtype=torch.float32).pin_memory()

    def memcpy_thread():
        dst = pinned.numpy()
        while not stop_t.is_set():
            np.copyto(dst, src)
    run_thread(memcpy_thread, "memcpy thread (-> pinned)")

    pageable = np.empty_like(src)

    def memcpy_pageable():
        while not stop_t.is_set():
            np.copyto(pageable, src)
    run_thread(memcpy_pageable, "memcpy thread (pageable)")

    def h2d_thread():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop_t.is_set():
                pinned.to("cuda", non_blocking=True)
                s.synchronize()
    run_thread(h2d_thread, "h2d thread (105 MB copies)")

    def gil_thread():
        x = 0
        while not stop_t.is_set():
            for _ in range(100000):
                x += 1
    run_thread(gil_thread, "gil thread (pure Python)")

    def fork_thread():
        # what a DataLoader does when it starts its workers: fork() of THIS (HIP-initialised, GBs mapped, pinned memory) process
        while not stop_t.is_set():
            pid = os.fork()
            if pid == 0:
                os._exit(0)
            os.waitpid(pid, 0)
            time.sleep(0.05)
    run_thread(fork_thread, "fork() + exit every 50 ms")

    def vfork_thread():
        import subprocess
        while not stop_t.is_set():
            subprocess.run(["/bin/true"])
            time.sleep(0.05)
    run_thread(vfork_thread, "subprocess /bin/true / 50 ms")

    def one_fork_then_idle():
        pid = os.fork()
        if pid == 0:
            time.sleep(4.0)
            os._exit(0)
        while not stop_t.is_set():
            time.sleep(0.01)
        os.waitpid(pid, 0)
    run_thread(one_fork_then_idle, "one forked child alive")

    stop = ctx.Event()
    procs = [ctx.Process(target=memcpy_process, args=(stop,), daemon=True) for _ in range(4)]
    for p in procs:
        p.start()
    time.sleep(3.0)
    print("%-28s  %7.2f / %7.2f" % (("memcpy processes 4",) + decode_loop()), flush=True)
    stop.set()
    for p in procs:
        p.join()


if __name__ == "__main__":
    main()
