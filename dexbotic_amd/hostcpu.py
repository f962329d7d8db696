"""Host-CPU budget of the process: the cgroup's CPU-bandwidth quota, and torch's intra-op thread pool held inside it.

Why this exists (round 6, profiles/r06_process_frame_tail.txt): torch sizes its OpenMP pool by the host's logical CPUs (128 threads
on the 256-CPU MI355X host) while the container's cgroup allows 16 CPUs per 100 ms CFS period (cpu.max "1600000 100000").  ONE host
op above ATen's parallel grain — the 393 KB torch.stack of two 256 x 256 frames in DexboticForCausalLM.process_images — wakes the
whole pool, whose workers then spin: 26 s of CPU time in 1.7 s of wall time, 15 of 18 periods throttled, and the kernel parks EVERY
thread of the cgroup for the rest of the period — among them the one inside hipGraphLaunch feeding the request's 440 kernels to the
queue.  The rocprofv3 trace shows it as ONE 22 - 65 ms hole between two arbitrary kernels of the graph with every kernel duration
unchanged: POST /process_frame p50 22.3 ms, p90 55.5 ms, every third request.  With the pool inside the quota: p90 22.4 ms.

The serving entry point (serve.InferenceServer) calls limit_host_threads() once; training goes through it as well
(trainer.NativeTrainer.__init__) because the feeder's pinned staging copies are ATen-parallel too.  Nothing here touches the GPU path.
"""
from __future__ import annotations

import math
import os
from typing import Optional


CGROUP_ROOT = "/sys/fs/cgroup"


def cpu_quota(root: Optional[str] = None) -> Optional[float]:
    """CPUs per scheduling period the cgroup of this process may use (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us);
    None when unlimited or unreadable"""
    root = root or CGROUP_ROOT
    try:
        p = os.path.join(root, "cpu.max")
        if os.path.exists(p):
            quota, period = open(p).read().split()[:2]
            return None if quota == "max" else float(quota) / float(period)
        q, per = os.path.join(root, "cpu/cpu.cfs_quota_us"), os.path.join(root, "cpu/cpu.cfs_period_us")
        if os.path.exists(q) and os.path.exists(per):
            quota, period = float(open(q).read()), float(open(per).read())
            return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        pass
    return None


def usable_cpus() -> int:
    """what this process can actually run on at once: min(affinity mask, cgroup quota rounded down), at least 1"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q is not None:
        n = min(n, max(1, int(math.floor(q))))
    return max(1, n)


def throttle_stats() -> dict:
    """the cgroup's CPU-bandwidth counters (cpu.stat): nr_periods, nr_throttled, throttled_usec, usage_usec — differences of two
    readings say whether a timed region ran into the quota"""
    out = {}
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            if os.path.exists(f):
                for ln in open(f):
                    k, v = ln.split()
                    if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
                        out[k] = int(v)
                break
        except (OSError, ValueError):
            pass
    return out


def limit_host_threads(reserve: int = 2, cap: Optional[int] = None, share: Optional[int] = None) -> int:
    """torch's intra-op pool <= usable_cpus() / share - reserve (the launching thread and the HIP runtime's own threads need CPU time
    inside the same quota; ``share`` = processes of this job inside the same cgroup, default LOCAL_WORLD_SIZE), never raised above its
    current size; ``cap`` bounds it further.  Env DXA_HOST_THREADS=n overrides (0 = leave torch alone).  Returns the pool size in force."""
    import torch
    env = os.environ.get("DXA_HOST_THREADS")
    if env is not None:
        if int(env) > 0:
            torch.set_num_threads(int(env))
        return torch.get_num_threads()
    if share is None:
        share = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    n = max(1, usable_cpus() // max(1, share) - reserve)
    if cap is not None:
        n = min(n, cap)
    if n < torch.get_num_threads():
        torch.set_num_threads(n)
    return torch.get_num_threads()


def upload(a, device, non_blocking: bool = True):
    """small host array (numpy or CPU tensor) -> ``device`` through PINNED memory with a non-blocking copy.  A pageable source
    (``torch.from_numpy(x).to(dev)``) makes the copy wait until the stream has drained: in the middle of a forward the launching
    thread loses everything it is ahead of the GPU, and what follows runs host-bound (profiles/r06_host_uploads.txt: MemVLA's step
    359 -> 323 ms for two 128-byte uploads).  The pinned block comes from torch's caching host allocator, which keeps it alive
    until the copy has run."""
    import numpy as np
    import torch
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    if torch.device(device).type == "cuda" and not t.is_pinned():
        t = t.pin_memory()
    return t.to(device, non_blocking=non_blocking)
