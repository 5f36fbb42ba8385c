"""Saver: named tensors <-> ``model.ckpt-<step>`` bundles plus the ``checkpoint`` state file.

Mirrors what ``tf.train.Supervisor``'s default ``tf.train.Saver()`` does for
/root/reference/distributed.py:110-111: all global variables
(``global_step`` int32 scalar, ``hid_w``, ``hid_b``, ``sm_w``, ``sm_b``) go into
``<dir>/model.ckpt-<global_step>.{index,data-00000-of-00001}``; the text file
``<dir>/checkpoint`` names the newest bundle and the (at most five) kept ones.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Optional

import numpy as np

from . import bundle

CKPT_BASENAME = "model.ckpt"
STATE_FILE = "checkpoint"


def _quote(s: str) -> str:
    return '"' + s.replace("\\", "\\\\").replace('"', '\\"') + '"'


def read_checkpoint_state(directory: str) -> Optional[Dict[str, List[str]]]:
    path = os.path.join(directory, STATE_FILE)
    if not os.path.exists(path):
        return None
    latest, allp = None, []
    with open(path) as f:
        for line in f:
            m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"((?:[^"\\]|\\.)*)"', line)
            if not m:
                continue
            val = m.group(2).replace('\\"', '"').replace("\\\\", "\\")
            if m.group(1) == "model_checkpoint_path":
                latest = val
            else:
                allp.append(val)
    if latest is None:
        return None
    return {"model_checkpoint_path": latest, "all_model_checkpoint_paths": allp}


def latest_checkpoint(directory: str) -> Optional[str]:
    """Prefix of the newest checkpoint in ``directory`` (``tf.train.latest_checkpoint``)."""
    st = read_checkpoint_state(directory)
    if st is None:
        return None
    p = st["model_checkpoint_path"]
    if not os.path.isabs(p):
        p = os.path.join(directory, p)
    return p if os.path.exists(p + ".index") else None


class Saver:
    def __init__(self, max_to_keep: int = 5):
        self.max_to_keep = max_to_keep
        self._kept: List[str] = []

    def save(self, tensors: Dict[str, np.ndarray], directory: str, global_step: int) -> str:
        os.makedirs(directory, exist_ok=True)
        if not self._kept:
            # a restarted job continues the list of its predecessor: old bundles stay listed and are retired in turn
            prev = read_checkpoint_state(directory)
            if prev is not None:
                self._kept = [p for p in prev["all_model_checkpoint_paths"] if os.path.exists(os.path.join(directory, p + ".index"))]
        name = f"{CKPT_BASENAME}-{int(global_step)}"
        prefix = os.path.join(directory, name)
        arrays = {}
        for k, v in tensors.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            arrays[k] = a.astype(np.int32) if k == "global_step" else a
        bundle.write_bundle(prefix, arrays)
        if name in self._kept:
            self._kept.remove(name)
        self._kept.append(name)
        while self.max_to_keep and len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            for suffix in (".index", ".data-00000-of-00001", ".meta"):
                try:
                    os.remove(os.path.join(directory, old + suffix))
                except FileNotFoundError:
                    pass
        tmp = os.path.join(directory, STATE_FILE + ".tmp")
        with open(tmp, "w") as f:
            f.write(f"model_checkpoint_path: {_quote(name)}\n")
            for k in self._kept:
                f.write(f"all_model_checkpoint_paths: {_quote(k)}\n")
        os.replace(tmp, os.path.join(directory, STATE_FILE))
        return prefix

    @staticmethod
    def restore(prefix: str) -> Dict[str, np.ndarray]:
        return bundle.read_bundle(prefix)
