"""Golden batches of the MemVLA episode sampler, produced by the reference's own classes
(dexbotic/exp/mem_trainer.py: _EpisodeScheduleBuilder / EpisodeBatchSampler) in this container.

    PYTHONPATH=/root/reference python -m oracle.gen_golden_sampler      # writes tests/golden/episode_sampler.npz

The reference module imports dexbotic.exp.trainer (-> loguru, megfile, deepspeed ...), none of which the sampler
uses: that one import is satisfied with an empty stand-in module.  Test infrastructure only."""
import os
import sys
import types

import numpy as np


def synthetic_index(seed: int = 5):
    """(dataset, file, frame) triples of 3 datasets x several episodes of uneven length (some shorter than a group),
    in a scrambled global order"""
    rs = np.random.RandomState(seed)
    rows = []
    for di in range(3):
        for fi in range(int(rs.randint(4, 9))):
            for fidx in range(int(rs.randint(2, 23))):
                rows.append((di, fi, fidx))
    rows = [rows[i] for i in rs.permutation(len(rows))]
    return np.array(rows, dtype=np.int64)


CASES = [  # (batch, group, seed, predict_length, world)
    (16, 4, 42, 0, 1), (16, 4, 42, 0, 2), (16, 8, 7, 3, 4), (4, 4, 1, 0, 8), (8, 8, 3, 16, 2)]


def main():
    stub = types.ModuleType("dexbotic.exp.trainer")
    stub.DexboticTrainer = type("DexboticTrainer", (), {})
    import dexbotic  # noqa: F401  (reference package, PYTHONPATH=/root/reference)
    sys.modules["dexbotic.exp.trainer"] = stub
    import torch.utils.data as tud
    tud.Sampler.__init__ = lambda self, *a, **k: None   # torch >= 2.4 dropped Sampler.__init__(data_source); the reference (torch 2.2.2) passes it
    from dexbotic.exp import mem_trainer as R
    gi = synthetic_index()
    out = {"global_index": gi, "cases": np.array(CASES, dtype=np.int64)}
    for ci, (B, G, seed, predict, world) in enumerate(CASES):
        ds = types.SimpleNamespace(global_index=[tuple(r) for r in gi.tolist()],
                                   action_process_func=types.SimpleNamespace(predict_length=predict))
        for rank in range(world):
            os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
            smp = R.EpisodeBatchSampler(ds, "group", B, G, seed=seed)
            for epoch in range(2):                      # iterating advances the epoch: seed + epoch
                batches = list(iter(smp))
                out[f"c{ci}_r{rank}_e{epoch}"] = np.array(batches, dtype=np.int64).reshape(-1, B)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "episode_sampler.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items() if k.startswith("c0")})


if __name__ == "__main__":
    main()
