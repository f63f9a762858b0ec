"""Pack / unpack the compiled-model blob defined by ``include/myosim_model.h``.

The C header is the single source of truth for section ids, dtypes and widths;
this module parses its X-macro table so the Python packer can never drift from
what the fp64 oracle and the HIP engine read.

Reference boundary this replaces: ``MjSpec.from_file(path).compile()`` returning
an ``MjModel`` (myosuite/envs/env_base.py:72,105).
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "myosim_model.h"))


def _parse_header(path: str = HEADER_PATH):
    with open(path, "r") as f:
        src = f.read()
    secs: List[Tuple[str, str, int]] = []
    for m in re.finditer(r"MM_SEC\(\s*([A-Z0-9_]+)\s*,\s*'([if])'\s*,\s*(\d+)\s*\)", src):
        secs.append((m.group(1), m.group(2), int(m.group(3))))
    consts: Dict[str, int] = {}
    for m in re.finditer(r"#define\s+(MM_[A-Z_]+)\s+(0x[0-9A-Fa-f]+|\d+)\s", src):
        consts[m.group(1)] = int(m.group(2), 0)
    # enums: sequential, optional "= value"
    for em in re.finditer(r"enum\s*\{([^}]*)\}", src, re.S):
        body = re.sub(r"/\*.*?\*/", "", em.group(1), flags=re.S)
        if "MM_SEC_ENUM" in body:
            continue
        val = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                val = int(v, 0)
            else:
                name = item
                val += 1
            consts[name] = val
    return secs, consts


SECTIONS, C = _parse_header()
SEC_INDEX = {name: i for i, (name, _, _) in enumerate(SECTIONS)}
NSEC = len(SECTIONS)
HEADER_WORDS = C["MM_HEADER_WORDS"]
MAGIC = C["MM_MAGIC"]
VERSION = C["MM_VERSION"]


def pack(arrays: Dict[str, np.ndarray]) -> np.ndarray:
    """arrays: section NAME (upper case) -> numpy array.  Missing sections are empty."""
    unknown = set(arrays) - set(SEC_INDEX)
    if unknown:
        raise KeyError(f"unknown model sections: {sorted(unknown)}")
    table_words = 2 * NSEC
    off = HEADER_WORDS + table_words
    chunks = []
    table = np.zeros(table_words, dtype=np.uint32)
    for i, (name, dt, width) in enumerate(SECTIONS):
        a = arrays.get(name)
        if a is None:
            a = np.zeros(0, dtype=np.int32 if dt == "i" else np.float32)
        a = np.ascontiguousarray(a)
        if dt == "i":
            if not np.issubdtype(a.dtype, np.integer):
                raise TypeError(f"section {name} must be integer, got {a.dtype}")
            w = a.astype(np.int32).reshape(-1).view(np.uint32)
        else:
            w = a.astype(np.float32).reshape(-1).view(np.uint32)
        if w.size % width:
            raise ValueError(f"section {name}: {w.size} words not a multiple of width {width}")
        table[2 * i] = off
        table[2 * i + 1] = w.size
        chunks.append(w)
        off += w.size
    head = np.array([MAGIC, VERSION, NSEC, off], dtype=np.uint32)
    return np.concatenate([head, table] + chunks).astype(np.uint32)


def unpack(blob: np.ndarray) -> Dict[str, np.ndarray]:
    blob = np.ascontiguousarray(blob, dtype=np.uint32)
    if blob[0] != MAGIC or blob[1] != VERSION or blob[2] != NSEC:
        raise ValueError("bad model blob header")
    out = {}
    for i, (name, dt, width) in enumerate(SECTIONS):
        off = int(blob[HEADER_WORDS + 2 * i])
        n = int(blob[HEADER_WORDS + 2 * i + 1])
        w = blob[off:off + n]
        a = w.view(np.int32) if dt == "i" else w.view(np.float32)
        out[name] = a.reshape(-1, width) if width > 1 else a.copy()
    return out
