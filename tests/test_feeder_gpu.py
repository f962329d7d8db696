"""data/feeder.DeviceFeeder on the GPU: batches staged in pinned memory and uploaded on a copy stream one step ahead arrive
intact, integer inputs stay on the host, and a fine-tune run fed that way (fresh token ids every step: a new splice plan with its
packed pinned upload per step) is bit-identical to the same run fed with device-resident batches."""
import numpy as np
import pytest
import torch

from tests.helpers import build_product, load_golden
from tests.test_parity_gpu import _batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _host_batches(g, n):
    out = []
    for i in range(n):
        rs = np.random.RandomState(300 + i)
        ids = g["input_ids"].copy()
        ids[:, 2:] = rs.randint(10, 400, size=ids[:, 2:].shape)            # fresh instruction tokens (the placeholder stays)
        b = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(g["attention_mask"].copy()),
                 labels=torch.from_numpy(ids.copy()),
                 images=torch.from_numpy(rs.standard_normal(g["images"].shape).astype(np.float32)),
                 actions=torch.from_numpy(rs.uniform(-1, 1, size=g["actions"].shape).astype(np.float32)),
                 noise=torch.from_numpy(rs.standard_normal(g["noise"].shape).astype(np.float32)),
                 timesteps=torch.from_numpy(g["timesteps"].copy()), drop_ids=torch.from_numpy(g["drop_u"] < 0.1))
        out.append(b)
    return out


def test_feeder_delivers_intact_batches_one_step_ahead(golden_dir):
    from dexbotic_amd.data.feeder import DeviceFeeder
    g, cfg, w = load_golden(golden_dir, "t1")
    host = _host_batches(g, 7)
    busy = torch.randn(64 << 20, device=DEV)
    got = []
    for b in DeviceFeeder(iter(host), DEV):
        for _ in range(20):
            busy.mul_(1.0001)                                              # keep the compute stream busy while the next upload runs
        got.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    assert len(got) == len(host)
    for h, d in zip(host, got):
        for k in ("input_ids", "attention_mask", "labels"):
            assert not d[k].is_cuda and torch.equal(d[k], h[k])           # integer inputs stay on the host
        for k in ("images", "actions", "noise"):
            assert d[k].is_cuda and torch.equal(d[k].cpu(), h[k]), k


def test_training_through_the_feeder_equals_device_resident_batches(golden_dir):
    from dexbotic_amd.data.feeder import DeviceFeeder
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    g, cfg, w = load_golden(golden_dir, "t1")
    host = _host_batches(g, 5)
    runs = {}
    for mode in ("device", "feeder"):
        m = build_product(cfg, w, "bfloat16", DEV, train=True)
        m.train()
        tr = NativeTrainer(m, OptimConfig(base_lr=1e-3, weight_decay=0.0, max_grad_norm=1.0))
        if mode == "device":
            src = ({k: v.to(DEV) for k, v in b.items()} for b in host)
        else:
            src = DeviceFeeder(iter(host), DEV)
        losses = [tr.step(b).item() for b in src]
        torch.cuda.synchronize()
        runs[mode] = (losses, m.store.master.clone())
    assert runs["device"][0] == runs["feeder"][0]
    assert torch.equal(runs["device"][1], runs["feeder"][1])
