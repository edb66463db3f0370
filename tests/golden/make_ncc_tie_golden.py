"""Golden cases that SIT ON A TIE, from the COMPILED REFERENCE (oracle/_ref, build container only):

  dup_*      the overlap regions of both tiles are the same periodic pattern: several shifts correlate perfectly, the NCC map holds
             several entries equal to 1.0f and compute_MAX_ind's "first strict maximum" (compute_funcs.cu:1294-1305) decides;
  thr_*      bead pairs, found by a seeded search, where an entry of the peak's row or column lies within 1e-6 of
             widthThr * peak -- the `<= thr` scans of compute_NCC_width (:160-282) are decided by the last bits;
  amax_*     noise pairs, found by a seeded search, whose two largest map entries are less than 1e-6 apart.

The search only SELECTS inputs (it measures margins on the reference's own maps); every stored expectation is what the reference
returned for them.  All tiles are stored (uint8).  Writes tests/golden/ncc_golden_ties.npz.
    python tests/golden/make_ncc_tie_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ncc_oracle as N  # noqa: E402

TILE = (28, 96, 96)
OVERLAP = 36
WANT_EACH = 3
F = np.float32


def quantise(t):
    q = np.clip(np.rint(t * 255.0), 0, 255).astype(np.uint8)
    return q, (q.astype(np.float32) / F(255.0)).astype(np.float32)


def periodic_pair(side, period, seed):
    """Both overlap views = one pattern with period `period` along the two in-plane axes; the rest of the tiles is noise."""
    rng = np.random.default_rng(seed)
    dk, di, dj = TILE
    base = rng.random((dk, period[0], period[1])).astype(np.float32)
    reps = (1, -(-di // period[0]), -(-dj // period[1]))
    pat = np.tile(base, reps)[:, :di, :dj]
    A = rng.random(TILE).astype(np.float32)
    B = rng.random(TILE).astype(np.float32)
    if side == 1:
        A[:, :, dj - OVERLAP:] = pat[:, :, :OVERLAP]
        B[:, :, :OVERLAP] = pat[:, :, :OVERLAP]
    else:
        A[:, di - OVERLAP:, :] = pat[:, :OVERLAP, :]
        B[:, :OVERLAP, :] = pat[:, :OVERLAP, :]
    return A, B


def width_margin(M, w_u, w_v, width_thr=F(0.80)):
    """smallest |M[c +- w] - thr| over the entries the two threshold scans of the row and the column through the centre visit"""
    H, W = M.shape
    cu, cv = H // 2, W // 2
    if np.isnan(M[cu, cv]) or int(np.nanargmax(M)) != cu * W + cv:
        return np.inf
    thr = F(width_thr * M[cu, cv])
    best = np.inf
    for line, c, rng_ in ((M[cu, :], cv, w_v), (M[:, cv], cu, w_u)):
        for sgn in (-1, 1):
            for w in range(1, rng_ + 1):
                v = line[c + sgn * w]
                best = min(best, abs(float(v) - float(thr)))
                if v <= thr:
                    break
    return best


MARGIN = 2e-6


def bead_pair(seed):
    A, B = N.tile_pair(TILE, OVERLAP, seed & 1, (0, 0, 0), seed=seed)
    return quantise(A)[1], quantise(B)[1]


def noise_pair(seed):
    rng = np.random.default_rng(seed)
    return quantise(rng.random(TILE, dtype=np.float32))[1], quantise(rng.random(TILE, dtype=np.float32))[1]


def probe_seed(args):
    seed, want_thr, want_amax = args
    side = seed & 1
    m_thr = m_amax = np.inf
    if want_thr:
        A, B = bead_pair(seed)
        r = N.pdalgo_execute(A, B, 8, 8, 1, side, OVERLAP, kind="ref", debug=True)
        di, dj, dk = r["delays"]
        m_thr = min(width_margin(r["maps"][0], di, dj), width_margin(r["maps"][1], di, dk), width_margin(r["maps"][2], dj, dk))
    if want_amax:
        A, B = noise_pair(seed)
        r = N.pdalgo_execute(A, B, 9, 9, 1, side, OVERLAP, kind="ref", debug=True)
        for mp_ in r["maps"]:
            v = np.sort(mp_[~np.isnan(mp_)].ravel())
            if v.size >= 2:
                m_amax = min(m_amax, float(v[-1]) - float(v[-2]))
    return seed, m_thr, m_amax


def main():
    assert N.have_ref(), "build oracle/_ref first: make -C oracle"
    cases = []   # (name, A, B, side, dmax)
    for i, (side, period) in enumerate([(1, (96, 4)), (0, (5, 96)), (1, (3, 5))]):
        A, B = periodic_pair(side, period, 50 + i)
        cases.append((f"dup_{i}", A, B, side, (6, 6, 1)))
    # seeded search on 8 processes, bounded: margins below MARGIN are inside the product's decision margin (4e-6)
    import multiprocessing as mp
    with mp.Pool(8) as pool:
        thr_hits, amax_hits, start = [], [], 1000
        while (len(thr_hits) < WANT_EACH or len(amax_hits) < WANT_EACH) and start < 40000:
            res = pool.map(probe_seed, [(sd, len(thr_hits) < WANT_EACH, len(amax_hits) < WANT_EACH) for sd in range(start, start + 800)])
            for sd, m_thr, m_amax in res:
                if m_thr < MARGIN and len(thr_hits) < WANT_EACH:
                    thr_hits.append(sd)
                    print("thr case: seed", sd, "margin", m_thr, flush=True)
                if m_amax < MARGIN and len(amax_hits) < WANT_EACH:
                    amax_hits.append(sd)
                    print("amax case: seed", sd, "gap", m_amax, flush=True)
            start += 800
            print("searched up to seed", start, flush=True)
    assert len(thr_hits) == WANT_EACH and len(amax_hits) == WANT_EACH, (thr_hits, amax_hits)
    for i, sd in enumerate(thr_hits):
        A, B = bead_pair(sd)
        cases.append((f"thr_{i}", A, B, sd & 1, (8, 8, 1)))
    for i, sd in enumerate(amax_hits):
        A, B = noise_pair(sd)
        cases.append((f"amax_{i}", A, B, sd & 1, (9, 9, 1)))
    out = {"names": np.array([c[0] for c in cases])}
    for name, A, B, side, dmax in cases:
        qa, A = quantise(A)
        qb, B = quantise(B)
        r = N.pdalgo_execute(A, B, dmax[0], dmax[1], dmax[2], side, OVERLAP, kind="ref", debug=True)
        assert r["rc"] == 0, name
        out[f"{name}/A_u8"], out[f"{name}/B_u8"] = qa, qb
        out[f"{name}/recipe"] = np.array([side, OVERLAP, *dmax], np.int64)
        out[f"{name}/coord"] = np.array(r["coord"], np.int32)
        out[f"{name}/NCC_maxs"] = r["NCC_maxs"]
        out[f"{name}/NCC_widths"] = np.array(r["NCC_widths"], np.int32)
        out[f"{name}/wRangeThr"] = np.array(r["wRangeThr"], np.int32)
        for m, nm in enumerate(["xy", "xz", "yz"]):
            out[f"{name}/map_{nm}"] = r["maps"][m]
        n_top = [int((mp == np.nanmax(mp)).sum()) if np.isfinite(np.nanmax(mp)) else 0 for mp in r["maps"]]
        print(f"{name:8s} coord={r['coord']} maxs={np.round(r['NCC_maxs'], 6)} widths={r['NCC_widths']} entries equal to the maximum={n_top}")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ncc_golden_ties.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
