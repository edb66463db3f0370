"""Rebuilds the input tiles of a golden NCC case (stored as uint8, or regenerated from the recipe and checked
against the stored SHA-256)."""
import hashlib
import importlib.util
import os

import numpy as np

_spec = importlib.util.spec_from_file_location(
    "make_ncc_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_ncc_golden.py"))
_gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gen)


def case_inputs(g, name):
    rec = [int(v) for v in g[f"{name}/recipe"]]
    tile, overlap, side, shift, seed, dmax = tuple(rec[0:3]), rec[3], rec[4], tuple(rec[5:8]), rec[8], tuple(rec[9:12])
    if f"{name}/A_u8" in g.files:
        qa, qb = g[f"{name}/A_u8"], g[f"{name}/B_u8"]
    else:
        A, B = _gen.make_tiles(tile, overlap, side, shift, seed, str(g[f"{name}/kind"]))
        qa, _ = _gen.quantise(A)
        qb, _ = _gen.quantise(B)
    sha = hashlib.sha256(qa.tobytes() + qb.tobytes()).hexdigest()
    assert sha == str(g[f"{name}/sha"]), f"synthetic generator drifted for golden case {name}"
    A = (qa.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    B = (qb.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    return A, B, overlap, side, dmax
