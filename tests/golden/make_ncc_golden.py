"""Generates tests/golden/ncc_golden.npz from the COMPILED REFERENCE (oracle/_ref/libcrossmips_ref.so,
built by oracle/Makefile from /root/reference/TeraStitcher/src/crossmips, unmodified).

Run in the build container only (the reference does not exist on the GPU box):
    python tests/golden/make_ncc_golden.py
Each case stores the generator recipe, a SHA-256 of the generated tiles (so drift of the synthetic
generator is detected), the tiles themselves as uint8 (value = q/255, like an 8-bit TIFF read through
TeraStitcher's loadImageStack: tiff2D.cpp:606-610) for the small cases, and the reference outputs:
9 scalars of NCC_descr_t, the mutated wRangeThr_*, the clamped delays, the six MIPs and three NCC maps.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ncc_oracle as N  # noqa: E402

# name, tile (k,i,j), overlap, side, shift (V,H,D), seed, displ_max (V,H,D), kind
CASES = [
    ("we_small", (32, 128, 128), 32, 1, (3, -2, 1), 7, (10, 10, 5), "beads"),
    ("ns_small", (32, 128, 128), 32, 0, (-4, 5, 0), 8, (10, 10, 5), "beads"),
    ("we_zero_shift", (32, 128, 128), 40, 1, (0, 0, 0), 9, (10, 10, 5), "beads"),
    ("ns_wide_overlap", (40, 160, 144), 64, 0, (6, -7, 2), 10, (12, 12, 8), "beads"),
    ("we_recentre", (40, 144, 160), 64, 1, (11, -9, 3), 11, (12, 12, 8), "beads"),      # peak near map border -> window moves
    ("we_out_of_range", (32, 128, 160), 60, 1, (14, 13, 0), 12, (8, 8, 5), "beads"),    # true shift outside the search range
    ("ns_thin_stack", (20, 128, 128), 48, 0, (2, 3, 1), 13, (10, 10, 10), "beads"),     # dimk < minDim_NCCsrc -> delayk = 0
    ("we_thin_27", (27, 128, 128), 48, 1, (-2, 1, 1), 14, (10, 10, 10), "beads"),       # delayk clamped to 2 (< minDim_NCCmap)
    ("we_all_zero", (26, 96, 96), 32, 1, (0, 0, 0), 15, (6, 6, 1), "zero"),             # 0/0 -> NaN maps
    ("ns_flat", (26, 96, 96), 32, 0, (0, 0, 0), 16, (6, 6, 1), "flat"),                 # constant tiles -> NaN maps
    ("we_noise", (30, 112, 112), 36, 1, (0, 0, 0), 17, (9, 9, 4), "noise"),             # uncorrelated noise: unreliable
    ("ns_not_tiled", (30, 100, 31), 40, 0, (1, -1, 0), 18, (8, 4, 4), "beads"),         # dimj < TILE_SIDE: plain-sum means
    ("we_c5_shape_small", (32, 256, 256), 38, 1, (4, -3, 1), 19, (25, 25, 10), "beads"),  # C5 parameters on a small tile
]
STORE_TILES = {"we_small", "ns_small", "ns_thin_stack", "we_noise", "ns_not_tiled"}


def make_tiles(tile, overlap, side, shift, seed, kind):
    if kind == "zero":
        return np.zeros(tile, np.float32), np.zeros(tile, np.float32)
    if kind == "flat":
        return np.full(tile, 0.25, np.float32), np.full(tile, 0.5, np.float32)
    if kind == "noise":
        rng = np.random.default_rng(seed)
        return rng.random(tile, dtype=np.float32), rng.random(tile, dtype=np.float32)
    A, B = N.tile_pair(tile, overlap, side, shift, seed)
    return A, B


def quantise(t):
    q = np.clip(np.rint(t * 255.0), 0, 255).astype(np.uint8)
    return q, (q.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def main():
    assert N.have_ref(), "build oracle/_ref first: make -C oracle"
    out = {"names": np.array([c[0] for c in CASES])}
    for name, tile, overlap, side, shift, seed, dmax, kind in CASES:
        A, B = make_tiles(tile, overlap, side, shift, seed, kind)
        qa, A = quantise(A)
        qb, B = quantise(B)
        r = N.pdalgo_execute(A, B, dmax[0], dmax[1], dmax[2], side, overlap, kind="ref", debug=True)
        assert r["rc"] == 0, name
        out[f"{name}/recipe"] = np.array([*tile, overlap, side, *shift, seed, *dmax], np.int64)
        out[f"{name}/kind"] = np.array(kind)
        out[f"{name}/sha"] = np.array(hashlib.sha256(qa.tobytes() + qb.tobytes()).hexdigest())
        if name in STORE_TILES:
            out[f"{name}/A_u8"], out[f"{name}/B_u8"] = qa, qb
        out[f"{name}/coord"] = np.array(r["coord"], np.int32)
        out[f"{name}/NCC_maxs"] = r["NCC_maxs"]
        out[f"{name}/NCC_widths"] = np.array(r["NCC_widths"], np.int32)
        out[f"{name}/wRangeThr"] = np.array(r["wRangeThr"], np.int32)
        out[f"{name}/INF_W"] = np.array(r["INF_W"], np.int32)
        out[f"{name}/delays_ijk"] = np.array(r["delays"], np.int32)
        for m, nm in enumerate(["xy1", "xz1", "yz1", "xy2", "xz2", "yz2"]):
            out[f"{name}/mip_{nm}"] = r["mips"][m]
        for m, nm in enumerate(["xy", "xz", "yz"]):
            out[f"{name}/map_{nm}"] = r["maps"][m]
        print(f"{name:22s} coord={r['coord']} maxs={np.round(r['NCC_maxs'], 4)} widths={r['NCC_widths']} "
              f"wR={r['wRangeThr']} delays={r['delays']}")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ncc_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
