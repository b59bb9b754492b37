"""Numeric form of the results of one chunk: what the GPU produced, before any Python item exists.

``FeaturePreprocessor.score_windows`` returns a ``WindowBatch``; ``items_from_batch`` turns it into
the reference's item dictionaries (feature_preprocessor.py:113-191).  A batch is also what travels
between the ranks of a sharded run: ``to_bytes`` is a short JSON header plus the raw arrays --
the fixed 24-byte window records and the CSR candidate lists of SURVEY.md section 8(e), and, when
configured, the f64 values of the ABBA-BABA family and DD -- so the multi-GPU gather moves one byte
row per rank (RCCL) instead of pickled dictionaries.
"""

from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from ..engine import RECORD_DTYPE, WindowResults

_MAGIC = b"SAIB1\0"


@dataclass
class ComboBatch:
    """One (ref, tgt, sources, outgroup) combination over the windows of the chunk."""

    ref_pop: str
    tgt_pop: str
    src_comb: tuple
    out_pop: Optional[str]
    windows: np.ndarray  # int64 [n_w][2] inclusive (start, end)
    nsnps: np.ndarray  # int32 [n_w]; 0 = the reference's "no site in this window" branch
    uq_names: list = field(default_factory=list)  # subset of ("U", "Q") in config order
    uq: Optional[WindowResults] = None  # records [len(uq_names)][n_w] + lists (entries = positions)
    four: Optional[np.ndarray] = None  # f64 [n_w][n_src][4]: fd, df, Danc, Dplus
    dd: Optional[np.ndarray] = None  # f64 [n_w][n_src]
    pos_dtype: str = "int32"


@dataclass
class WindowBatch:
    chr_name: str
    combos: list  # ComboBatch, in WindowGenerator.combinations() order

    # -- transport -------------------------------------------------------------------------

    def to_bytes(self) -> bytes:
        arrays: list[np.ndarray] = []

        def put(a) -> Optional[int]:
            if a is None:
                return None
            arrays.append(np.ascontiguousarray(a))
            return len(arrays) - 1

        combos = []
        for c in self.combos:
            meta = dict(ref=c.ref_pop, tgt=c.tgt_pop, src=list(c.src_comb), out=c.out_pop, uq_names=list(c.uq_names),
                        pos_dtype=c.pos_dtype, windows=put(c.windows), nsnps=put(c.nsnps), four=put(c.four), dd=put(c.dd))  # fmt: skip
            if c.uq is not None:
                uq = c.uq.separate_lists()  # rows that share lists (merged U / Q sets) get their own for the trip
                meta.update(rec=put(np.ascontiguousarray(uq.records).view(np.uint8).reshape(-1)), n_sets=int(uq.records.shape[0]),
                            cdd_u=put(uq.cdd_u), cdd_q=put(uq.cdd_q))  # fmt: skip
            combos.append(meta)
        head = json.dumps(
            dict(chr=self.chr_name, combos=combos,
                 arrays=[dict(dtype=a.dtype.str, shape=list(a.shape)) for a in arrays])  # fmt: skip
        ).encode()
        parts = [_MAGIC, struct.pack("<q", len(head)), head]
        parts.append(b"\0" * (-sum(map(len, parts)) % 8))  # payload starts 8-byte aligned
        for a in arrays:
            raw = a.tobytes()
            parts.append(raw)
            parts.append(b"\0" * (-len(raw) % 8))
        return b"".join(parts)

    @classmethod
    def from_bytes(cls, buf) -> "WindowBatch":
        buf = bytes(buf)
        if buf[: len(_MAGIC)] != _MAGIC:
            raise ValueError("not a WindowBatch byte row")
        o = len(_MAGIC)
        (n_head,) = struct.unpack_from("<q", buf, o)
        o += 8
        head = json.loads(buf[o : o + n_head].decode())
        o += n_head
        o += -o % 8
        arrays = []
        for spec in head["arrays"]:
            dt = np.dtype(spec["dtype"])
            n = int(np.prod(spec["shape"], dtype=np.int64)) * dt.itemsize
            arrays.append(np.frombuffer(buf, dtype=dt, count=n // dt.itemsize, offset=o).reshape(spec["shape"]).copy())
            o += n + (-n % 8)

        def get(i):
            return None if i is None else arrays[i]

        combos = []
        for m in head["combos"]:
            windows = get(m["windows"])
            uq = None
            if "rec" in m:
                n_w = int(windows.shape[0])
                rec = get(m["rec"]).view(RECORD_DTYPE).reshape(m["n_sets"], n_w)
                off = np.zeros((m["n_sets"], n_w, 2), dtype=np.int64)
                for k, name in enumerate(("u_count", "n_cdd_q")):
                    flat = rec[name].reshape(-1).astype(np.int64)
                    off[:, :, k] = (np.cumsum(flat) - flat).reshape(m["n_sets"], n_w)
                cdd_u, cdd_q = get(m["cdd_u"]), get(m["cdd_q"])
                # a truncated or mismatched row must not become shifted candidate lists in the log files
                if int(rec["u_count"].sum()) != cdd_u.size or int(rec["n_cdd_q"].sum()) != cdd_q.size:
                    raise ValueError("window batch: candidate lists do not match the records' counts")
                uq = WindowResults(rec, off, cdd_u, cdd_q)
            n_w = int(windows.shape[0])
            for name, ndim in (("nsnps", 1), ("four", 3), ("dd", 2)):
                arr = get(m[name])
                if arr is not None and (arr.ndim != ndim or arr.shape[0] != n_w):
                    raise ValueError(f"window batch: '{name}' of shape {arr.shape} for {n_w} windows")
            combos.append(
                ComboBatch(m["ref"], m["tgt"], tuple(m["src"]), m["out"], windows, get(m["nsnps"]), list(m["uq_names"]), uq,
                           get(m["four"]), get(m["dd"]), m["pos_dtype"])  # fmt: skip
            )
        return cls(head["chr"], combos)
