#!/usr/bin/env python
"""Golden vectors for the input pipeline, produced by the REFERENCE's own classes over the
live Pillow / transformers of this container:

    dexbotic/data/dataset/rgb_preprocess.py  PreprocessRGB            (imported from /root/reference)
    dexbotic/tokenization/tokenization.py    tokenizer_image_token    (imported from /root/reference)
    dexbotic/data/collator.py                DataCollatorForSupervisedDataset

    python oracle/gen_golden_image.py        ->  tests/golden/image_t1.npz

Frames are regenerated from seeds by oracle.image_oracle.synthetic_image, so only outputs are stored."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import image_oracle as IO  # noqa: E402

REF = "/root/reference"
CASES = [  # (name, h, w, aspect, pad_mode)
    ("vga_pad", 480, 640, "pad", "mean"), ("tall_pad", 640, 480, "pad", "mean"), ("hd_pad", 720, 1280, "pad", "mean"),
    ("vga_zero", 480, 640, "pad", "zero"), ("same", 224, 224, "pad", "mean"), ("up_pad", 180, 200, "pad", "mean"),
    ("sq300", 300, 300, "pad", "mean"), ("vga_crop", 480, 640, None, "mean"), ("tall_crop", 500, 375, None, "mean"),
    ("odd_pad", 333, 517, "pad", "mean"), ("tiny", 17, 40, "pad", "mean"),
]


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeTok:
    """word-hash tokenizer with a BOS, enough for tokenizer_image_token / the collator"""
    bos_token_id, eos_token_id, pad_token_id, model_max_length = 1, 2, 0, 24

    def __call__(self, text):
        r = types.SimpleNamespace()
        r.input_ids = [self.bos_token_id] + [3 + (sum(map(ord, w)) % 997) for w in text.split()]
        return r


def main():
    from transformers import CLIPImageProcessor
    rgb = load(os.path.join(REF, "dexbotic/data/dataset/rgb_preprocess.py"), "ref_rgb")
    proc = CLIPImageProcessor()
    out = {"pil_version": np.array(Image.__version__ if hasattr(Image, "__version__") else __import__("PIL").__version__),
           "image_mean": np.asarray(proc.image_mean, np.float64), "image_std": np.asarray(proc.image_std, np.float64)}
    names = []
    for i, (name, h, w, aspect, pad_mode) in enumerate(CASES):
        frame = IO.synthetic_image(h, w, 100 + i)
        pre = rgb.PreprocessRGB(proc, image_aspect_ratio=aspect, image_pad_mode=pad_mode)
        pv = pre(Image.fromarray(frame)).numpy()
        assert pv.shape == (3, 224, 224) and pv.dtype == np.float32
        # the uint8 stage on its own, straight from Pillow
        pil = Image.fromarray(frame)
        if aspect == "pad":
            pil = rgb.PreprocessRGB.expand2square(pil, IO.pad_color(pad_mode, proc.image_mean))
        oh, ow = IO.resize_output_size(pil.size[1], pil.size[0], 224)
        u8 = np.asarray(pil.resize((ow, oh), resample=Image.BICUBIC))
        u8 = IO.center_crop(u8, 224, 224)
        out[f"{name}/u8"] = u8
        out[f"{name}/meta"] = np.array([h, w, 100 + i, 1 if aspect == "pad" else 0, 1 if pad_mode == "zero" else 0], np.int64)
        if i < 3 or name in ("vga_crop", "tiny"):
            out[f"{name}/pixel_values"] = pv
        out[f"{name}/pv_sum"] = np.array(pv.astype(np.float64).sum())
        out[f"{name}/pv_abs"] = np.array(np.abs(pv.astype(np.float64)).sum())
        names.append(name)
    out["cases"] = np.array(names)

    # ---- prompt side
    tk_mod_src = open(os.path.join(REF, "dexbotic/tokenization/tokenization.py")).read()
    # the module imports the conversation templates of the package; only tokenizer_image_token is needed here
    ns = {"torch": torch, "IMAGE_TOKEN_INDEX": -200, "IGNORE_INDEX": -100}
    start = tk_mod_src.index("def tokenizer_image_token")
    end = tk_mod_src.index("def tokenize_dexbotic")
    exec(compile(tk_mod_src[start:end], "ref_tokenization", "exec"), ns)      # runs the reference's function body as is
    tok = FakeTok()
    prompts = ["<image>\npick up the red block", "look <image> and <image> then move left", "no image here", "<image>"]
    for j, pr in enumerate(prompts):
        out[f"prompt{j}/ids"] = np.asarray(ns["tokenizer_image_token"](pr, tok), np.int64)
    out["prompts"] = np.array(prompts)
    stub = types.ModuleType("dexbotic.constants")
    stub.IGNORE_INDEX = -100
    sys.modules.setdefault("dexbotic", types.ModuleType("dexbotic"))
    sys.modules["dexbotic.constants"] = stub
    col = load(os.path.join(REF, "dexbotic/data/collator.py"), "ref_collator")
    for tag, pad in (("pad0", 0), ("padeos", 2)):
        tok2 = FakeTok()
        tok2.pad_token_id = pad
        rows = [torch.tensor([1, 5, 2, 9, 0, 7]), torch.tensor([1, 8, 2]), torch.tensor(list(range(1, 31)))]
        labs = [torch.tensor([-100, -100, 2, 9, 0, 7]), torch.tensor([-100, 8, 2]), torch.tensor(list(range(1, 31)))]
        b = col.DataCollatorForSupervisedDataset(tok2)([{"input_ids": r.clone(), "labels": l.clone()} for r, l in zip(rows, labs)])
        out[f"collate_{tag}/input_ids"] = b["input_ids"].numpy()
        out[f"collate_{tag}/labels"] = b["labels"].numpy()
        out[f"collate_{tag}/attention_mask"] = b["attention_mask"].numpy()
    conv_mod = load(os.path.join(REF, "dexbotic/tokenization/conversation.py"), "ref_conversation")
    conv_prompts = []
    for name in ("dexbotic", "step", "llama_3"):
        for stub in (" ", None):
            conv = conv_mod.conv_templates[name].copy()
            conv.append_message(conv.roles[0], "<image>\n" + "pick up the red block")
            conv.append_message(conv.roles[1], stub)
            conv_prompts.append(conv.get_prompt())
        conv = conv_mod.conv_templates[name].copy()
        conv.append_message(conv.roles[0], ("what is <image> this", None, "Pad"))
        conv.append_message(conv.roles[1], "a cube")
        conv.append_message(conv.roles[0], "and now?")
        conv_prompts.append(conv.get_prompt())
    out["conv_prompts"] = np.array(conv_prompts)
    path = os.path.join(ROOT, "tests", "golden", "image_t1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
