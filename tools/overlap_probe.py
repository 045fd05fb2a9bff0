#!/usr/bin/env python3
"""tools/overlap_probe.py [--config c2] [--reps 3]

Co-residency probe: does the memory-latency-bound prefilter (E1-E4) run UNDER the VALU-bound gapped stage (E5/E6)
when both are on the device at once?  Two engines on ONE device hold the same database; engine A runs `prefilter`, engine B runs `align`
on its (already installed) hit lists.  Timed: each alone, then both started together from two host threads (ctypes releases the GIL).
    t_both ~= max(tP, tA)  -> the stages overlap: restructuring the step into query batches pays
    t_both ~= tP + tA      -> they do not: the device serialises them (CU occupancy / LDS), nothing to gain
Also: the same with the prefilter cut into query halves (what a pipelined step would launch).  Prints one JSON line."""
import argparse, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UC_ALLOW_SYNTHETIC", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
import unicore_amd as U


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--work", default="/tmp/uc_bench")
    ap.add_argument("--priority", action="store_true", help="prefilter engine on high-priority HIP streams, gapped-stage engine on low-priority ones")
    a = ap.parse_args()
    prot, fam, scale, seed, opts, _ = bench.CONFIGS[a.config]
    db = bench.gen_db(os.path.join(a.work, a.config), prot, fam, scale, seed)
    if a.priority:                 # engine A (prefilter) on high-priority streams, engine B (gapped stage) on low-priority ones
        os.environ["UC_STREAM_PRIORITY"] = "high"
    A = U.Engine(opts, threads=4)
    if a.priority:
        os.environ["UC_STREAM_PRIORITY"] = "low"
    B = U.Engine(opts, threads=4)
    os.environ.pop("UC_STREAM_PRIORITY", None)
    A.load_db(db); B.load_db(db)
    n = A.n
    for _ in range(2):
        A.prefilter(); B.prefilter(); B.align()
    out = {"config": a.config, "sequences": n, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "stream_priorities": bool(a.priority)}
    out["prefilter_alone_s"] = timed(lambda: A.prefilter(), a.reps)
    out["align_alone_s"] = timed(lambda: B.align(), a.reps)
    out["prefilter_halves_alone_s"] = timed(lambda: (A.prefilter(0, n, 0, n // 2), A.prefilter(0, n, n // 2, n)), a.reps)

    def both(pre):
        ends = {}
        def run(name, fn):
            fn(); ends[name] = time.perf_counter()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=("prefilter", pre)), threading.Thread(target=run, args=("align", lambda: B.align()))]
        for t in th: t.start()
        for t in th: t.join()
        return {"wall": time.perf_counter() - t0, "prefilter_done": ends["prefilter"] - t0, "align_done": ends["align"] - t0}

    out["both"] = [both(lambda: A.prefilter()) for _ in range(a.reps)]
    out["both_halves"] = [both(lambda: (A.prefilter(0, n, 0, n // 2), A.prefilter(0, n, n // 2, n))) for _ in range(a.reps)]
    tp, ta = min(out["prefilter_alone_s"]), min(out["align_alone_s"])
    tb = min(r["wall"] for r in out["both"])
    out["summary"] = {"tP": tp, "tA": ta, "sum": tp + ta, "max": max(tp, ta), "t_both": tb,
                      "hidden_fraction_of_prefilter": (tp + ta - tb) / tp}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
