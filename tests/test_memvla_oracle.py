"""Pin the CPU restatement of MemVLA (oracle/memvla_oracle.py) against golden vectors from the live reference
MemVLAForCausalLM (oracle/gen_golden_memvla.py): a 'group'-mode training batch (2 episodes x 3 frames, token-merge
consolidation active) and a 4-frame inference episode.  CPU only."""
import os

import numpy as np
import torch

from oracle import cogact_oracle as O
from oracle import memvla_oracle as M
from oracle.weights import make_weights, weights_crc


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def load(golden_dir):
    g = np.load(os.path.join(golden_dir, "memvla_t1.npz"), allow_pickle=False)
    cfg = O.OracleConfig()
    w = make_weights(M.memvla_shapes(cfg, int(g["per_token_size"])), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    return g, cfg, w


def test_memvla_training_step_matches_reference(golden_dir):
    g, cfg, w = load(golden_dir)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    bank = M.MemBank(int(g["mem_length"]))
    out = M.memvla_forward(sd, cfg, bank, t(g["input_ids"]), t(g["attention_mask"]), t(g["images"]), t(g["actions"]),
                           [list(map(int, r)) for r in g["indexes"]], t(g["noise"]), t(g["timesteps"]), t(g["drop_u"]) < 0.1)
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    out["loss"].backward()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel(sd[key[5:]].grad.numpy(), g[key]) < 5e-5, key
    gsq = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.grad is not None)
    assert abs(gsq ** 0.5 - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])


def test_memvla_inference_episode_matches_reference(golden_dir):
    g, cfg, w = load(golden_dir)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    t = torch.from_numpy
    norms = {"min": [-1.0] * cfg.action_dim, "max": [1.0] * cfg.action_dim}
    bank = M.MemBank(int(g["mem_length"]))
    with torch.no_grad():
        for f in range(g["infer_frames"].shape[0]):
            if f == 0:
                bank.reset()
            acts, _ = M.memvla_inference_action(sd, cfg, bank, f, t(g["infer_prompt"]), t(g["infer_frames"][f:f + 1]),
                                                t(g["infer_inits"][f]), norms)
            assert rel(acts, g["infer_actions"][f]) < 2e-5, f


def test_memvla_training_step_with_retrieval_dropout_matches_reference(golden_dir):
    """the reference's retrieval blocks are built with dropout 0.1 (SDPA weights + two nn.Dropout in the FFN); golden
    memvla_drop_t1.npz = the same batch through the reference with that dropout ON and every mask injected (MaskFeed)"""
    from oracle.gen_golden_memvla import MaskFeed
    g, cfg, w = load(golden_dir)
    gd = np.load(os.path.join(golden_dir, "memvla_drop_t1.npz"), allow_pickle=False)
    assert int(gd["weights_crc"]) == int(g["weights_crc"])
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    feed = MaskFeed(int(gd["mask_seed"]), float(gd["p_drop"]))
    bank = M.MemBank(int(g["mem_length"]), mask_fn=feed)
    out = M.memvla_forward(sd, cfg, bank, t(g["input_ids"]), t(g["attention_mask"]), t(g["images"]), t(g["actions"]),
                           [list(map(int, r)) for r in g["indexes"]], t(g["noise"]), t(g["timesteps"]), t(g["drop_u"]) < 0.1)
    assert feed.k == int(gd["masks_drawn"])
    assert abs(float(out["loss"]) - float(gd["loss"])) < 1e-5 * abs(float(gd["loss"]))
    assert abs(float(gd["loss"]) - float(g["loss"])) > 1e-4          # the masks do change the result
    out["loss"].backward()
    for key in gd.files:
        if key.startswith("grad/"):
            assert rel(sd[key[5:]].grad.numpy(), gd[key]) < 5e-5, key
    gsq = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.grad is not None)
    assert abs(gsq ** 0.5 - float(gd["grad_norm"])) < 1e-4 * float(gd["grad_norm"])
