import os, sys, time
sys.path.insert(0, "/root/repo")
import bench, unicore_amd as U
wd = "/tmp/uc_bench/p50"; bench.gen_db(wd, 50, 6000, 1.0, 0x5EED0002); db = os.path.join(wd, "db")
wq = "/tmp/uc_bench/p5q"; bench.gen_db(wq, 5, 6000, 1.0, 0x5EED0002); qdb = os.path.join(wq, "db")
U.search(qdb, db, "/tmp/uc_bench/w_aln", "/tmp/uc_bench/tmp", "-c 0.8")
t = time.perf_counter(); st = U.search(qdb, db, "/tmp/uc_bench/w_aln", "/tmp/uc_bench/tmp", "-c 0.8"); dt = time.perf_counter() - t
print("wall %.3f" % dt, dict(zip(U.STAGES, [round(x, 3) for x in st["stage_seconds"]])), "sw_ms %.0f pre_ms %.0f" % (st["sw_kernel_ms"], st["prefilter_kernel_ms"]))
