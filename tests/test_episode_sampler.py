"""MemVLA episode-group sampler (dexbotic_amd/exp/mem_trainer.py) against batches produced by the reference's own
EpisodeBatchSampler (tests/golden/episode_sampler.npz, oracle/gen_golden_sampler.py): integer work, bit-exact; plus the
data-parallel properties the rank sharding exists for."""
import os
import types

import numpy as np
import pytest

from dexbotic_amd.exp.mem_trainer import CollatePassThrough, EpisodeBatchSampler, build_group_batches, episode_map


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "episode_sampler.npz"))


def _ds(gi, predict):
    return types.SimpleNamespace(global_index=[tuple(r) for r in gi.tolist()],
                                 action_process_func=types.SimpleNamespace(predict_length=predict))


def test_batches_equal_reference(gold, monkeypatch):
    gi = gold["global_index"]
    for ci, (B, G, seed, predict, world) in enumerate(gold["cases"].tolist()):
        for rank in range(world):
            monkeypatch.setenv("RANK", str(rank))
            monkeypatch.setenv("WORLD_SIZE", str(world))
            smp = EpisodeBatchSampler(_ds(gi, predict), "group", B, G, seed=seed)
            for epoch in range(2):
                got = np.array(list(iter(smp)), dtype=np.int64).reshape(-1, B)
                assert np.array_equal(got, gold[f"c{ci}_r{rank}_e{epoch}"]), (ci, rank, epoch)


def test_ranks_never_share_an_episode_and_groups_are_ordered(gold):
    gi = gold["global_index"]
    world, B, G = 4, 16, 4
    seen = []
    for rank in range(world):
        batches = build_group_batches(gi, B, G, seed=9, epoch=3, rank=rank, world=world)
        assert batches and all(len(b) == B for b in batches)
        eps = set()
        for b in batches:
            for j in range(0, B, G):                       # each group of G = one episode, frames ascending
                grp = gi[b[j:j + G]]
                assert len({(int(r[0]), int(r[1])) for r in grp}) == 1
                assert np.all(np.diff(grp[:, 2]) >= 0)
                eps.add((int(grp[0, 0]), int(grp[0, 1])))
        seen.append(eps)
    for i in range(world):
        for j in range(i + 1, world):
            assert not (seen[i] & seen[j])
    # the shards partition the shuffled episode list: together they can reach every episode
    all_eps = {k for k, _ in episode_map(gi)}
    assert set().union(*seen) <= all_eps


def test_edge_cases():
    assert build_group_batches(np.zeros((0, 3), np.int64), 8, 4, 0, 0, 0, 1) == []
    gi = np.array([(0, 0, f) for f in range(3)], dtype=np.int64)          # one episode shorter than the group: padded
    (b,) = build_group_batches(gi, 4, 4, 0, 0, 0, 1)
    assert b == [0, 1, 2, 2]
    assert build_group_batches(gi, 4, 4, 0, 0, rank=1, world=2) == []      # rank without episodes
    with pytest.raises(ValueError):
        build_group_batches(gi, 6, 4, 0, 0, 0, 1)
    with pytest.raises(NotImplementedError):
        EpisodeBatchSampler(_ds(gi, 0), "stream", 4, 4)


def test_collate_pass_through():
    c = CollatePassThrough(lambda batch: {"n": len(batch), "keys": sorted(batch[0])})
    out = c([{"x": 1, "indexes": (0, 1, 2)}, {"x": 2, "indexes": (0, 1, 3)}])
    assert out == {"n": 2, "keys": ["x"], "indexes": [(0, 1, 2), (0, 1, 3)]}
