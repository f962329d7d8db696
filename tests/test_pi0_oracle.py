"""Pin the CPU restatement of the pi0 policy (oracle/pi0_oracle.py) against golden vectors produced by the live
reference Pi0ForCausalLM (oracle/gen_golden_pi0.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import pi0_oracle as P
from oracle.weights import make_weights, weights_crc


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def load(golden_dir):
    g = np.load(os.path.join(golden_dir, "pi0_t1.npz"), allow_pickle=False)
    cfg = P.Pi0OracleConfig()
    w = make_weights(P.pi0_shapes(cfg), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    return g, cfg, w


def test_pi0_forward_loss_and_grads_match_reference(golden_dir):
    g, cfg, w = load(golden_dir)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    out = P.pi0_forward(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["images"]), t(g["image_masks"]),
                        t(g["states"]), t(g["actions"]), t(g["noise"]), t(g["time"]))
    assert rel(out["v_t"].detach().numpy(), g["v_t"]) < 2e-5
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    out["loss"].backward()
    for key in g.files:
        if key.startswith("grad/"):
            assert rel(sd[key[5:]].grad.numpy(), g[key]) < 5e-5, key
        elif key.startswith("gradN/"):
            assert sd[key[6:]].grad is not None, key
    gsq = sum(float(v.grad.double().pow(2).sum()) for v in sd.values() if v.grad is not None)
    assert abs(gsq ** 0.5 - float(g["grad_norm"])) < 1e-4 * float(g["grad_norm"])


def test_pi0_inference_matches_reference(golden_dir):
    g, cfg, w = load(golden_dir)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    t = torch.from_numpy
    with torch.no_grad():
        acts = P.pi0_inference_action(sd, cfg, t(g["input_ids"]), t(g["attention_mask"]), t(g["states"]), t(g["images"]),
                                      t(g["image_masks"]), t(g["init_noise"]), 10)
    assert rel(acts.numpy(), g["infer_actions"]) < 2e-5


def test_pi0_oracle_matches_reference_classes_at_real_width(golden_dir):
    """tests/golden/pi0_real_ref.npz: the reference Pi0ForCausalLM at Gemma-2B / SigLIP-So400m widths, 2 layers, chunk 50
    (oracle/gen_golden_pi0_real.py) — training step (fp32) and 10-step fp32 inference"""
    import os
    import zlib

    import numpy as np
    import torch

    from oracle import gen_golden_pi0_real as PR
    from oracle import pi0_oracle as P
    from oracle.weights import make_weights, weights_crc
    g = np.load(os.path.join(golden_dir, "pi0_real_ref.npz"), allow_pickle=False)
    x = PR.inputs()
    assert zlib.crc32(x["images"].tobytes()) == int(g["images_crc"])
    w = make_weights(P.pi0_shapes(PR.REAL), int(g["seed"]))
    assert weights_crc(w) == int(g["weights_crc"])
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in w.items()}
    t = torch.from_numpy
    o = P.pi0_forward(sd, PR.REAL, t(x["input_ids"]), t(x["attention_mask"]), t(x["images"]), t(x["image_masks"]),
                      t(x["states"]), t(x["actions"]), t(x["noise"]), t(x["time"]))
    o["loss"].backward()
    got = PR.summarize({n: p.grad for n, p in sd.items()}, o["loss"].item(), o["v_t"].detach().numpy())
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() /
                             (np.abs(np.asarray(b, np.float64)).max() + 1e-12))
    for k, v in got.items():
        assert rel(v, g["fp32/" + k]) < 5e-5, (k, rel(v, g["fp32/" + k]))
    with torch.no_grad():
        acts = P.pi0_inference_action(sd, PR.REAL, t(x["input_ids"]), t(x["attention_mask"]), t(x["states"]), t(x["images"]),
                                      t(x["image_masks"]), t(x["init_noise"]), 10)
    assert rel(acts.numpy(), g["fp32/infer_actions"]) < 5e-5
