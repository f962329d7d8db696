#!/usr/bin/env python
"""The tail of POST /process_frame under a trace: N requests through bench.process_frame_latency on the
full-size DB-CogACT, to be run under

    rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -f csv -d gpurun_out/pf -o pf -- python scripts/pf_trace.py 24

and read with scripts/pf_trace_report.py.  Prints the per-request wall latencies (host) so that the report can be matched to them.

    python scripts/pf_trace.py N [mode ...]      one model, one process_frame_latency(N) leg per mode:
        plain       the server loop as bench.py runs it
        sleepX      X ms of host sleep before every request (is the stall tied to wall time or to the request count?)
        keepX       a one-thread spin kernel of X ms on a side stream after every response (the chip never idles between requests)
        back2back   inference_action only, device-resident inputs (the loop that has no tail)
        threadsN    torch.set_num_threads(N) for the leg (the host work of a request is a PNG decode and a 393 KB torch.stack)
Every leg is bracketed by the cgroup's CPU-bandwidth counters (cpu.stat: nr_throttled / throttled_usec) and the process's thread count.
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    modes = sys.argv[2:] or ["plain"]
    dev = torch.device("cuda", 0)
    args = types.SimpleNamespace(llm_layers=int(os.environ.get("LLM_LAYERS", "28")), vit_layers=24, dtype="bfloat16")
    model, cfg, _, _ = bench.build_model(args, dev)
    model.eval()
    os.environ["DXA_BENCH_PF_DUMP"] = "1"
    import dexbotic_amd.serve as S
    orig = S.InferenceServer.get_response
    side = torch.cuda.Stream(device=dev)
    # cycles of torch.cuda._sleep per ms (its counter is not the shader clock)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(10_000_000)
    e1.record()
    torch.cuda.synchronize()
    cyc_per_ms = 10_000_000 / e0.elapsed_time(e1)
    print(f"# torch.cuda._sleep: {cyc_per_ms:.0f} cycles per ms")
    def cg():
        out = {}
        for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
            if os.path.exists(f):
                for ln in open(f):
                    k, v = ln.split()
                    if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
                        out[k] = int(v)
                break
        return out
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(f):
            print(f"# {f}: {open(f).read().strip()}")
    import psutil
    print(f"# os.cpu_count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, torch threads {torch.get_num_threads()}, "
          f"process threads {psutil.Process().num_threads()}")
    nt0 = torch.get_num_threads()
    for mode in modes:
        torch.set_num_threads(int(mode[7:]) if mode.startswith("threads") else nt0)
        c0 = cg()
        sleep_ms = float(mode[5:]) if mode.startswith("sleep") else 0.0
        keep_ms = float(mode[4:]) if mode.startswith("keep") else 0.0

        def patched(self, text, images, _s=sleep_ms, _k=keep_ms):
            if _s > 0:
                time.sleep(_s * 1e-3)
            r = orig(self, text, images)
            if _k > 0:
                with torch.cuda.stream(side):
                    torch.cuda._sleep(int(_k * cyc_per_ms))
            return r
        S.InferenceServer.get_response = patched
        t0 = time.perf_counter()
        if mode == "back2back":
            reqs = [bench.synthetic_batch(1, 2, 32, dev, seed=7 + 13 * i) for i in range(8)]
            norms = {"min": [-1.0] * 7, "max": [1.0] * 7}
            lat = []
            for i in range(6 + n_req):
                b1 = reqs[i % len(reqs)]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                model.inference_action(b1["input_ids"], b1["images"], {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms})
                lat.append(round(1e3 * (time.perf_counter() - t1), 1))
            out = {"p50_ms": float(np.median(lat[6:])), "p90_ms": float(np.percentile(lat[6:], 90)), "max_ms": max(lat[6:]), "all_ms": lat[6:]}
        else:
            out = bench.process_frame_latency(model, n_req=n_req)
        print(f"# mode {mode}: {n_req} requests in {time.perf_counter() - t0:.2f} s (warm-up included)")
        c1 = cg()
        print(json.dumps({"mode": mode, "cgroup_delta": {k: c1[k] - c0.get(k, 0) for k in c1}, "process_threads": psutil.Process().num_threads(),
                          **{k: v for k, v in out.items() if k not in ("stage_ms", "inner_ms")}}), flush=True)
    S.InferenceServer.get_response = orig


if __name__ == "__main__":
    main()
