"""The action-inference wire format: POST /process_frame, multipart form with a `text` field and one or more `image`
PNG files, answered with {"response": [[...7 floats...] x chunk]} — dexbotic/exp/base_exp.py:638-653 (route),
dexbotic/exp/cogact_exp.py:146-177 (_get_response), dexbotic/client.py:33-61 (the caller).

Only PNG decoding and prompt formatting stay on the host; the frames go to the MI355X as uint8 and are padded,
resized, cropped and normalised there (DexboticForCausalLM.process_images -> dxa_image_preprocess)."""
from __future__ import annotations

import io
import os
import time
from typing import Callable, List, Optional

import torch

from .constants import DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
from .tokenization import conversation as conversation_lib
from .tokenization.tokenization import tokenizer_image_token


class InferenceServer:
    """model: anything with process_images / inference_action / device / dtype / config.chat_template"""

    def __init__(self, model, tokenizer, norm_stats: Optional[dict] = None, port: int = 7891, cfg_scale: float = 1.5,
                 num_ddim_steps: int = 10, assistant_stub: Optional[str] = " ", log: Optional[Callable[[str], None]] = None):
        self.model, self.tokenizer, self.norm_stats, self.port = model, tokenizer, norm_stats, port
        self.inference_args = {"cfg_scale": cfg_scale, "num_ddim_steps": num_ddim_steps, "action_norms": norm_stats}
        self.assistant_stub = assistant_stub                     # cogact_exp.py:159 appends ' ', base_exp.py:683 None
        self.log = log
        self.last_ms = None
        # the host side of a request is one PNG decode and a few hundred KB of byte shuffling: torch's intra-op pool (sized by the
        # host's logical CPUs) is held inside the container's CPU quota, or its spinning workers get the whole process throttled
        # in the middle of a request (hostcpu.py; POST /process_frame p90 55 -> 22 ms)
        from .hostcpu import limit_host_threads
        self.host_threads = limit_host_threads(cap=8)
        self.stage_ms = {} if os.environ.get("DXA_SERVE_STAGES") else None

    # ---- one request ------------------------------------------------------------------------------------------
    def build_prompt(self, text: str) -> str:
        template = getattr(self.model.config, "chat_template", "dexbotic")
        conv = conversation_lib.conv_templates[template].copy()
        conv.append_message(conv.roles[0], DEFAULT_IMAGE_TOKEN + "\n" + text)
        conv.append_message(conv.roles[1], self.assistant_stub)
        return conv.get_prompt()

    def get_response(self, text: str, images: List) -> list:
        """images: file-like objects / paths holding PNG (or anything PIL opens).  cogact_exp.py:146-177"""
        from PIL import Image
        t0 = time.monotonic()
        stages = self.stage_ms if self.stage_ms is not None else None      # tuning: per-stage times (a device sync per stage)

        def mark(name, t_prev):
            if stages is None:
                return t_prev
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t = time.monotonic()
            stages.setdefault(name, []).append(round(1e3 * (t - t_prev), 2))
            return t
        frames = [Image.open(f).convert("RGB") for f in images]
        t = mark("png_decode", t0)
        pix = self.model.process_images(frames).to(dtype=self.model.dtype)
        if len(frames) > 1:
            pix = pix.unsqueeze(0)                               # [1, views, 3, H, W]
        t = mark("process_images", t)
        ids = tokenizer_image_token(self.build_prompt(text), self.tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt")
        ids = ids.unsqueeze(0).to(self.model.device)
        t = mark("prompt", t)
        with torch.no_grad():
            out = self.model.inference_action(ids, pix, dict(self.inference_args))
        out = out.tolist() if hasattr(out, "tolist") else out
        mark("inference_action", t)
        self.last_ms = 1e3 * (time.monotonic() - t0)
        if self.log:
            self.log(f"process_frame: {len(frames)} view(s), {self.last_ms:.1f} ms")
        return out

    def inference_single(self, image_path, prompt: str) -> list:
        """the exp scripts' ``--task inference_single`` (playground/benchmarks/libero/libero_cogact.py:70-72):
        ``_get_response(prompt, [image_path])`` on one image file (or a list of view files)"""
        paths = [image_path] if isinstance(image_path, (str, bytes)) or hasattr(image_path, "read") else list(image_path)
        return self.get_response(prompt, paths)

    # ---- Flask plumbing ---------------------------------------------------------------------------------------
    def create_app(self):
        from flask import Flask, jsonify, request
        app = Flask(__name__)

        def process_frame():
            files = [io.BytesIO(f.read()) for f in request.files.getlist("image")]
            return jsonify({"response": self.get_response(request.form.get("text"), files)})

        app.add_url_rule("/process_frame", "process_frame", process_frame, methods=["POST"])
        return app

    def run(self) -> None:
        self.create_app().run(host="0.0.0.0", port=self.port, debug=False, threaded=False)


def encode_png(frame) -> bytes:
    """RGB uint8 [h, w, 3] -> PNG bytes, what dexbotic/client.py:40-44 sends (there via cv2.imencode)"""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(frame).save(buf, format="PNG")
    return buf.getvalue()
