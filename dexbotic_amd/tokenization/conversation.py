"""Prompt templates: dexbotic/tokenization/conversation.py:52-233 (host string formatting; the three templates the
reference registers, with identical text so that the token ids — and hence the splice plan — are identical)."""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional, Sequence

_ASSISTANT_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
                     "The assistant gives helpful, detailed, and polite answers to the user's questions.")


class SeparatorStyle(Enum):
    TWO = auto()
    PLAIN = auto()
    LLAMA_3 = auto()


def _text(message):
    return message[0] if isinstance(message, tuple) else message


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Sequence[str]
    messages: List[List[Optional[str]]]
    offset: int
    sep_style: SeparatorStyle
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, version=self.version)

    def get_prompt(self) -> str:
        msgs = list(self.messages)
        if msgs and isinstance(msgs[0][1], tuple):            # (text, image, mode): the image tag leads the first turn
            role, first = msgs[0]
            msgs[0] = (role, "<image>\n" + first[0].replace("<image>", "").strip())
        seps = (self.sep, self.sep2)
        if self.sep_style == SeparatorStyle.TWO:
            out = self.system + seps[0]
            for i, (role, m) in enumerate(msgs):
                out += f"{role}: {_text(m)}{seps[i % 2]}" if m else f"{role}:"
            return out
        if self.sep_style == SeparatorStyle.PLAIN:
            return self.system + "".join(_text(m) + seps[i % 2] for i, (_, m) in enumerate(msgs) if m)
        if self.sep_style == SeparatorStyle.LLAMA_3:
            out = self.system + self.sep
            for i, (role, m) in enumerate(msgs):
                out += role + _text(m) + (self.sep if i < len(msgs) - 1 else self.sep2) if m else role
            return out
        raise ValueError(f"Invalid style: {self.sep_style}")


conv_templates = {
    "dexbotic": Conversation(system=_ASSISTANT_SYSTEM, roles=("USER", "ASSISTANT"), version="dexbotic", messages=[], offset=0,
                             sep_style=SeparatorStyle.TWO, sep=" ", sep2="<|endoftext|>"),
    "step": Conversation(system=_ASSISTANT_SYSTEM, roles=("USER", "ASSISTANT"), version="step", messages=[], offset=0,
                         sep_style=SeparatorStyle.TWO, sep=" ", sep2="<|im_end|>"),
    "llama_3": Conversation(
        system="<|begin_of_text|><|start_header_id|>system<|end_header_id|>\n\nYou are a helpful language and vision assistant. "
               "You are able to understand the visual content that the user provides, "
               "and assist the user with a variety of tasks using natural language.",
        roles=("<|start_header_id|>user<|end_header_id|>\n\n", "<|start_header_id|>assistant<|end_header_id|>\n\n"),
        version="llama_v3", messages=[], offset=0, sep_style=SeparatorStyle.LLAMA_3, sep="<|eot_id|>", sep2="<|end_of_text|>"),
}
