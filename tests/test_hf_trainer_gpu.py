"""SURVEY.md section 8(b) acceptance on the GPU: the native policy under the loop the reference's exp scripts run —
``DexboticTrainer(transformers.Trainer)`` (dexbotic/exp/trainer.py:18-138: HF ``training_step`` per micro-batch, global-norm clip,
``optimizer.step()``, ``lr_scheduler.step()``, ``model.zero_grad()``) — and the ``inference_single`` task
(playground/benchmarks/libero/libero_cogact.py:70-72 -> cogact_exp.py:146-177).

  * plain HF ``Trainer`` + the optimizer it creates itself (torch.optim.AdamW over the arena views): the native model keeps its
    bf16 shadows fresh and its gradient arena attached without anybody telling it (ParamStore.external_prelude / w());
  * ``NativeDexboticTrainer`` (the opt-in replacement for DexboticTrainer): same loop, fused arena optimizer;
  * both against ``NativeTrainer`` on identical batches: loss trajectories within 1e-3.
"""
import os
import tempfile

import numpy as np
import pytest
import torch

from tests.helpers import build_product, load_golden
from tests.test_parity_gpu import _batch

pytestmark = pytest.mark.gpu
DEV = "cuda"
LR, STEPS = 1e-3, 3


def _batches(g, n):
    """n distinct deterministic batches derived from the golden one (other actions / noise; same token plan)"""
    out = []
    for i in range(n):
        b = _batch(g)
        rs = np.random.RandomState(100 + i)
        b["actions"] = torch.from_numpy(rs.uniform(-1, 1, size=tuple(b["actions"].shape)).astype(np.float32)).to(DEV)
        b["noise"] = torch.from_numpy(rs.standard_normal(tuple(b["noise"].shape)).astype(np.float32)).to(DEV)
        out.append(b)
    return out


def _native_losses(g, cfg, w, dtype, batches, accum=1):
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.trainer import NativeTrainer
    m = build_product(cfg, w, dtype, DEV, train=True)
    m.train()
    tr = NativeTrainer(m, OptimConfig(base_lr=LR, weight_decay=0.0, max_grad_norm=1.0), grad_accum=accum)
    losses = [tr.step(b).item() for b in batches]
    torch.cuda.synchronize()
    return losses, m.store.master.clone()


def _args(**kw):
    from transformers import TrainingArguments
    base = dict(output_dir=tempfile.mkdtemp(), per_device_train_batch_size=2, gradient_accumulation_steps=1, max_steps=STEPS,
                report_to=[], learning_rate=LR, weight_decay=0.0, max_grad_norm=1.0, remove_unused_columns=False,
                save_strategy="no", logging_steps=1, lr_scheduler_type="constant", bf16=True, dataloader_num_workers=0)
    base.update(kw)
    return TrainingArguments(**base)


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_plain_hf_trainer_and_torch_adamw_train_the_native_model(golden_dir, dtype):
    """HF Trainer.training_step -> clip_grad_norm_ -> (torch.optim.AdamW).step() -> lr_scheduler.step() -> model.zero_grad(),
    three times, on the native model; zero_grad(set_to_none=True) by the optimizer thrown in.  The bf16 run only works if the
    shadows follow the fp32 masters torch's AdamW moves."""
    from transformers import Trainer
    g, cfg, w = load_golden(golden_dir, "t1")
    batches = _batches(g, STEPS)
    want, want_master = _native_losses(g, cfg, w, dtype, batches)
    m = build_product(cfg, w, dtype, DEV, train=True)
    tr = Trainer(model=m, args=_args(bf16=(dtype == "bfloat16")), train_dataset=[0] * 8)
    tr.create_optimizer_and_scheduler(num_training_steps=STEPS)
    assert isinstance(tr.optimizer, torch.optim.AdamW)
    tr.current_gradient_accumulation_steps = 1
    got = []
    m.zero_grad()
    for i, b in enumerate(batches):
        got.append(float(tr.training_step(m, b)))
        params = [p for p in m.parameters() if p.grad is not None]
        assert len(params) > 10
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        tr.optimizer.step()
        tr.lr_scheduler.step()
        if i == 1:
            tr.optimizer.zero_grad(set_to_none=True)          # user code does this; the next forward re-attaches the arena views
        else:
            m.zero_grad()
    torch.cuda.synchronize()
    print(dtype, "native", want, "hf+torch.optim", got)
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-3 * abs(b), (got, want)
    assert got[0] == want[0]
    # the masters moved as far as with the fused optimizer (same AdamW, other arithmetic order)
    moved = (m.store.master - want_master).abs().max().item()
    assert moved <= 6.5 * LR, moved      # (sign-like first steps on ~0 gradients may go either way)


def test_native_dexbotic_trainer_steps_and_full_train_loop(golden_dir):
    """the opt-in DexboticTrainer replacement: its training_step / ArenaAdamW.step / scheduler / zero_grad sequence equals
    NativeTrainer bit for bit (same machinery, HF's loop on top), with gradient accumulation 2; then trainer.train() runs HF's
    whole inner loop (dataloader, collator, scheduler, logging) for three optimizer steps."""
    from dexbotic_amd.exp.config import ExpConfig, OptimizerConfig, TrainerConfig
    from dexbotic_amd.exp.trainer import ArenaAdamW, NativeDexboticTrainer, link_exp_config
    g, cfg, w = load_golden(golden_dir, "t1")
    batches = _batches(g, 2 * STEPS)
    want, want_master = _native_losses(g, cfg, w, "bfloat16", batches, accum=2)
    exp = ExpConfig(TrainerConfig(output_dir=tempfile.mkdtemp(), num_train_steps=STEPS, per_device_train_batch_size=2,
                                  gradient_accumulation_steps=2, logging_steps=1, dataloader_num_workers=0,
                                  lr_scheduler_type="constant", save_strategy="no"),
                    OptimizerConfig(base_lr=LR, weight_decay=0.0))
    args = link_exp_config(exp, report_to=[])
    assert args.max_grad_norm == 0.0 and not args.gradient_checkpointing and args.deepspeed is None
    m = build_product(cfg, w, "bfloat16", DEV, train=True)
    # pass by pass here (the bit-for-bit comparison with NativeTrainer's two passes); the coalesced default: the test below
    tr = NativeDexboticTrainer(model=m, args=args, train_dataset=[0] * 8, exp_config=exp, native={"coalesce_micro_batches": False})
    tr.create_optimizer_and_scheduler(num_training_steps=STEPS)
    assert isinstance(tr.optimizer, ArenaAdamW) and not tr.core.coalesce
    tr.current_gradient_accumulation_steps = 2
    got = []
    m.zero_grad()
    for i, b in enumerate(batches):
        got.append(float(tr.training_step(m, b)) * 2)                # training_step reports loss / accumulation steps
        if i % 2 == 1:
            tr.optimizer.step()
            tr.lr_scheduler.step()
            m.zero_grad()
    torch.cuda.synchronize()
    assert np.allclose(got, want, rtol=1e-6, atol=0), (got, want)
    assert torch.equal(m.store.master, want_master)

    # HF's own loop end to end: dataset of single samples -> default collator -> 3 optimizer steps of 2 micro-batches
    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 16

        def __getitem__(self, i):
            rs = np.random.RandomState(i)
            return {"input_ids": torch.from_numpy(g["input_ids"][i % 2]), "attention_mask": torch.from_numpy(g["attention_mask"][i % 2]),
                    "labels": torch.from_numpy(g["input_ids"][i % 2]), "images": torch.from_numpy(g["images"][i % 2]),
                    "actions": torch.from_numpy(rs.uniform(-1, 1, size=g["actions"].shape[1:]).astype(np.float32))}
    m2 = build_product(cfg, w, "bfloat16", DEV, train=True)
    tr2 = NativeDexboticTrainer(model=m2, args=link_exp_config(exp, report_to=[]), train_dataset=DS(), exp_config=exp)
    before = m2.store.master.clone()
    out = tr2.train()
    torch.cuda.synchronize()
    assert out.global_step == STEPS and np.isfinite(out.training_loss)
    assert tr2.core.global_step == STEPS and tr2.core.opt.step_count == STEPS
    assert (m2.store.master - before).abs().max().item() > 0
    # integer inputs stayed on the host all the way into the model (no device->host copy for the splice plan)
    assert not tr2._prepare_inputs({"input_ids": torch.zeros(2, 4, dtype=torch.long), "images": torch.zeros(2, 3)})["input_ids"].is_cuda


def test_accumulation_group_coalesced_into_one_pass_equals_two_passes(golden_dir):
    """the reference recipe's two micro-batches per optimizer step (cogact_exp.py:41-46) run as ONE pass over the concatenated
    batch (NativeTrainer coalesce_micro_batches, NativeDexboticTrainer's default): HF's training_step sequence reports 0 for the
    held micro-batch and the group's loss sum for the last; losses, the clipped global norm and the parameters after three
    optimizer steps against the pass-by-pass run — same arithmetic up to the order of fp32 sums over the batch rows.  Token rows
    of unequal length are right-padded to the longer micro-batch."""
    from dexbotic_amd.engine import OptimConfig
    from dexbotic_amd.exp.config import ExpConfig, OptimizerConfig, TrainerConfig
    from dexbotic_amd.exp.trainer import NativeDexboticTrainer, link_exp_config
    from dexbotic_amd.trainer import NativeTrainer, coalesce_batches
    g, cfg, w = load_golden(golden_dir, "t1")
    batches = _batches(g, 2 * STEPS)
    for dtype, tol in (("float32", 2e-5), ("bfloat16", 2e-3)):
        want, want_master = _native_losses(g, cfg, w, dtype, batches, accum=2)
        exp = ExpConfig(TrainerConfig(output_dir=tempfile.mkdtemp(), num_train_steps=STEPS, per_device_train_batch_size=2,
                                      gradient_accumulation_steps=2, logging_steps=1, dataloader_num_workers=0,
                                      lr_scheduler_type="constant", save_strategy="no", bf16=(dtype == "bfloat16")),
                        OptimizerConfig(base_lr=LR, weight_decay=0.0))
        m = build_product(cfg, w, dtype, DEV, train=True)
        tr = NativeDexboticTrainer(model=m, args=link_exp_config(exp, report_to=[]), train_dataset=[0] * 8, exp_config=exp)
        tr.create_optimizer_and_scheduler(num_training_steps=STEPS)
        assert tr.core.coalesce
        tr.current_gradient_accumulation_steps = 2
        got = []
        m.zero_grad()
        for i, b in enumerate(batches):
            got.append(float(tr.training_step(m, b)))
            if i % 2 == 1:
                tr.optimizer.step()
                tr.lr_scheduler.step()
                m.zero_grad()
        torch.cuda.synchronize()
        assert tr.core.coalesced_steps == STEPS and tr.core.global_step == STEPS
        print(dtype, "two passes", want, "coalesced", got)
        for k in range(STEPS):
            assert got[2 * k] == 0.0
            pair = 0.5 * (want[2 * k] + want[2 * k + 1])                      # mean of the two micro-batch losses
            assert abs(got[2 * k + 1] - pair) <= tol * abs(pair), (dtype, k, got, want)
        # Adam's first steps are sign-like: a gradient that is ~0 may step either way under another summation order (up to
        # 2 LR per step, as in the plain-HF test above); everywhere else the parameters agree to rounding
        diff = (m.store.master - want_master).abs()
        far = (diff > 0.5 * LR).float().mean().item()
        print(dtype, "parameters: max distance", diff.max().item(), "share further than LR/2 apart", far)
        assert diff.max().item() <= 6.5 * LR and far < 0.02, (dtype, diff.max().item(), far)
    # the injected draws ([R * B, ...], row r * B + b) are re-interleaved per repeat; ragged token rows are right-padded
    b0, b1 = _batch(g), _batch(g)
    B, R = b0["input_ids"].shape[0], b0["noise"].shape[0] // b0["input_ids"].shape[0]
    b1 = dict(b1, noise=b1["noise"] + 1.0)
    b1["input_ids"] = torch.nn.functional.pad(b1["input_ids"], (0, 3), value=5)
    b1["attention_mask"] = torch.nn.functional.pad(b1["attention_mask"], (0, 3), value=1)
    b1["labels"] = torch.nn.functional.pad(b1["labels"], (0, 3), value=5)
    mg = coalesce_batches([b0, b1])
    S = b1["input_ids"].shape[1]
    assert mg["input_ids"].shape == (2 * B, S) and mg["noise"].shape[0] == 2 * R * B
    assert torch.equal(mg["attention_mask"][:B, -3:], torch.zeros_like(mg["attention_mask"][:B, -3:]))
    assert torch.equal(mg["labels"][:B, -3:], torch.full_like(mg["labels"][:B, -3:], -100))
    for r in range(R):
        assert torch.equal(mg["noise"][r * 2 * B:r * 2 * B + B], b0["noise"][r * B:(r + 1) * B])
        assert torch.equal(mg["noise"][r * 2 * B + B:(r + 1) * 2 * B], b1["noise"][r * B:(r + 1) * B])
    assert coalesce_batches([b0, dict(b1, extra=torch.zeros(1))]) is None
    # ... and the merged ragged pair gives the loss the two batches give apart
    m = build_product(cfg, w, "float32", DEV, train=False)
    with torch.no_grad():
        l0, l1, lm = m(**b0).loss.item(), m(**b1).loss.item(), m(**mg).loss.item()
    assert abs(lm - 0.5 * (l0 + l1)) <= 2e-5 * abs(lm), (l0, l1, lm)


def test_native_dexbotic_trainer_short_last_group_and_checkpoints(golden_dir):
    """(a) an epoch whose length does not divide by gradient_accumulation_steps: HF closes the last group after ONE micro-batch
    and calls optimizer.step() (the reference's defaults: accumulation 2, drop_last False) — the native core must follow the
    group HF actually built; (b) save_steps = 1: checkpoint-N holds config.json + model.safetensors (+ norm_stats.json copied
    from the run directory, dexbotic/exp/trainer.py:38-82) and reloads through from_pretrained with identical weights."""
    import json
    from dexbotic_amd.exp.config import ExpConfig, OptimizerConfig, TrainerConfig
    from dexbotic_amd.exp.trainer import NativeDexboticTrainer, link_exp_config
    from dexbotic_amd.model.cogact.cogact_arch import CogACTForCausalLM
    g, cfg, w = load_golden(golden_dir, "t1")
    out_dir = tempfile.mkdtemp()
    with open(os.path.join(out_dir, "norm_stats.json"), "w") as f:
        json.dump({"min": [-1.0] * 7, "max": [1.0] * 7}, f)
    exp = ExpConfig(TrainerConfig(output_dir=out_dir, num_train_epochs=1, num_train_steps=-1, per_device_train_batch_size=2,
                                  gradient_accumulation_steps=2, logging_steps=1, dataloader_num_workers=0,
                                  lr_scheduler_type="constant", save_strategy="steps", save_steps=1, save_total_limit=None,
                                  save_only_model=True),
                    OptimizerConfig(base_lr=LR, weight_decay=0.0))

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 6                                   # 3 micro-batches of 2: groups of 2 + 1

        def __getitem__(self, i):
            rs = np.random.RandomState(i)
            return {"input_ids": torch.from_numpy(g["input_ids"][i % 2]), "attention_mask": torch.from_numpy(g["attention_mask"][i % 2]),
                    "labels": torch.from_numpy(g["input_ids"][i % 2]), "images": torch.from_numpy(g["images"][i % 2]),
                    "actions": torch.from_numpy(rs.uniform(-1, 1, size=g["actions"].shape[1:]).astype(np.float32))}
    m = build_product(cfg, w, "bfloat16", DEV, train=True)
    tr = NativeDexboticTrainer(model=m, args=link_exp_config(exp, report_to=[]), train_dataset=DS(), exp_config=exp)
    out = tr.train()
    torch.cuda.synchronize()
    assert out.global_step == 2 and tr.core.global_step == 2 and tr.core.opt.step_count == 2
    assert not tr.core.update_due and tr.core.micro % tr.core.grad_accum == 0
    for step in (1, 2):
        ck = os.path.join(out_dir, f"checkpoint-{step}")
        assert os.path.exists(os.path.join(ck, "config.json")) and os.path.exists(os.path.join(ck, "model.safetensors")), os.listdir(ck)
        assert os.path.exists(os.path.join(ck, "norm_stats.json"))
    back = CogACTForCausalLM.from_pretrained(os.path.join(out_dir, "checkpoint-2"), torch_dtype=torch.bfloat16, device=DEV, train=False)
    assert torch.equal(back.store.master, m.store.master)


def test_gradient_checkpointing_enable_recomputes_and_the_hf_loop_honours_it(golden_dir, monkeypatch):
    """the reference trains with gradient_checkpointing=True (base_exp.py:245; trainer.py:101,120 hand it to HF, which calls
    model.gradient_checkpointing_enable): on the native model that switch is ParamStore.recompute.  HF's whole loop with the
    switch forwarded (DEXBOTIC_AMD_GRAD_CHECKPOINTING=1) ends on the same parameters, bit for bit, as with resident activations."""
    from dexbotic_amd.exp.config import ExpConfig, OptimizerConfig, TrainerConfig
    from dexbotic_amd.exp.trainer import NativeDexboticTrainer, link_exp_config
    g, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, "float32", DEV, train=True)
    assert m.supports_gradient_checkpointing and not m.is_gradient_checkpointing
    m.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})
    assert m.store.recompute and m.is_gradient_checkpointing
    m.gradient_checkpointing_disable()
    assert not m.store.recompute
    monkeypatch.setenv("DEXBOTIC_AMD_ACCEPT_GRAD_CHECKPOINTING", "1")       # old opt-in: accept the call, stay resident
    with pytest.warns(UserWarning):
        m.gradient_checkpointing_enable()
    assert not m.store.recompute
    monkeypatch.delenv("DEXBOTIC_AMD_ACCEPT_GRAD_CHECKPOINTING")
    with pytest.raises(NotImplementedError):
        m.to(torch.bfloat16)
    assert m.to(DEV) is m and m.cuda() is m

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 16

        def __getitem__(self, i):
            rs = np.random.RandomState(i)
            return {"input_ids": torch.from_numpy(g["input_ids"][i % 2]), "attention_mask": torch.from_numpy(g["attention_mask"][i % 2]),
                    "labels": torch.from_numpy(g["input_ids"][i % 2]), "images": torch.from_numpy(g["images"][i % 2]),
                    "actions": torch.from_numpy(rs.uniform(-1, 1, size=g["actions"].shape[1:]).astype(np.float32))}
    exp = ExpConfig(TrainerConfig(output_dir=tempfile.mkdtemp(), num_train_steps=STEPS, per_device_train_batch_size=2,
                                  gradient_accumulation_steps=2, logging_steps=1, dataloader_num_workers=0,
                                  lr_scheduler_type="constant", save_strategy="no"),
                    OptimizerConfig(base_lr=LR, weight_decay=0.0))
    assert exp.trainer_config.gradient_checkpointing                       # the reference default
    finals = {}
    for tag in ("resident", "recompute"):
        if tag == "recompute":
            monkeypatch.setenv("DEXBOTIC_AMD_GRAD_CHECKPOINTING", "1")
        args = link_exp_config(exp, report_to=[], seed=7, data_seed=7)
        assert args.gradient_checkpointing == (tag == "recompute")
        m2 = build_product(cfg, w, "bfloat16", DEV, train=True)
        tr = NativeDexboticTrainer(model=m2, args=args, train_dataset=DS(), exp_config=exp)
        torch.manual_seed(11)                                              # the model draws noise / timesteps / drops itself
        out = tr.train()
        torch.cuda.synchronize()
        assert out.global_step == STEPS and m2.store.recompute == (tag == "recompute")
        finals[tag] = (m2.store.master.clone(), out.training_loss)
    assert finals["resident"][1] == finals["recompute"][1]
    assert torch.equal(finals["resident"][0], finals["recompute"][0])


def test_inference_single_on_an_image_file(golden_dir, tmp_path):
    """exp.inference_single(image_path, prompt) = InferenceConfig._get_response(prompt, [image_path]) (cogact_exp.py:146-177):
    PIL image from a file -> process_images -> .to(model.dtype) -> conversation prompt -> tokenizer_image_token ->
    inference_action, on a bf16 model as the reference loads it (cogact_exp.py:134-138)"""
    import types

    from PIL import Image

    from dexbotic_amd.serve import InferenceServer
    from dexbotic_amd.tokenization.tokenization import tokenizer_image_token
    from oracle import image_oracle as IO
    _, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, "bfloat16", DEV, train=False)
    m.eval()

    class Tok:
        bos_token_id = None

        def __call__(self, text):
            return types.SimpleNamespace(input_ids=[3 + (sum(map(ord, wd)) % (cfg.vocab_size - 3)) for wd in text.split()])
    norms = {"min": [-1.0] * 7, "max": [1.0] * 7}
    srv = InferenceServer(m, Tok(), norm_stats=norms)
    frame = IO.synthetic_image(96, 128, 5)
    path = str(tmp_path / "frame.png")
    Image.fromarray(frame).save(path)
    torch.manual_seed(3)
    acts = np.asarray(srv.inference_single(path, "What action should the robot take to pick up the bowl?"))
    assert acts.shape == (cfg.chunk_size, cfg.action_dim) and np.isfinite(acts).all() and np.abs(acts).max() <= 1.0 + 1e-6
    # the same pieces called by hand
    pix = m.process_images([Image.open(path).convert("RGB")]).to(dtype=m.dtype)
    ids = tokenizer_image_token(srv.build_prompt("What action should the robot take to pick up the bowl?"), Tok(),
                                return_tensors="pt")[None].to(DEV)
    torch.manual_seed(3)
    want = np.asarray(m.inference_action(ids, pix, {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms}))
    assert np.array_equal(acts, want)
