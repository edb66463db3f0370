"""CPU ORACLE worker (test infrastructure, NOT product code): MIP-NCC pairs through the compiled reference crossmips
(oracle/_ref/libcrossmips_ref.so) or the C restatement, in a process of its own -- the reference keeps static state
(libcrossmips.cpp:91, compute_funcs.cu:621-629), so pairs run in parallel as processes, like the reference's own MPI ranks
(Parastitcher.py:1440-1560).  bench_ncc.py starts one worker per host core BEFORE anything touches the GPU and feeds them
after its timed region.  Protocol, one JSON object per line on stdin:
    {"a": "<A.npy>", "b": "<B.npy>", "displ": [V, H, D], "direction": 0|1, "overlap": n, "kind": "ref"|"oracle"}
-> the worker loads the stacks and answers "ready"; the next line ("go") starts the pair; the answer is one JSON line
{"seconds": compute time, "coord", "NCC_maxs", "NCC_widths", "wRangeThr"}.  EOF ends the worker."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    np = ncc_oracle = None
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        if np is None:
            import numpy as np
            from oracle import ncc_oracle
        job = json.loads(line)
        A = np.ascontiguousarray(np.load(job["a"], mmap_mode="r"))
        B = np.ascontiguousarray(np.load(job["b"], mmap_mode="r"))
        sys.stdout.write("ready\n")
        sys.stdout.flush()
        sys.stdin.readline()            # all workers start their pair together
        t0 = time.perf_counter()
        r = ncc_oracle.pdalgo_execute(A, B, *job["displ"], job["direction"], job["overlap"], kind=job["kind"])
        dt = time.perf_counter() - t0
        print(json.dumps({"seconds": dt, "coord": [int(v) for v in r["coord"]], "NCC_maxs": [float(v) for v in r["NCC_maxs"]],
                          "NCC_widths": [int(v) for v in r["NCC_widths"]], "wRangeThr": [int(v) for v in r["wRangeThr"]]}), flush=True)
        del A, B


if __name__ == "__main__":
    main()
