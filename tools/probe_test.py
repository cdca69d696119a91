import sys, os, time
sys.path.insert(0, "/root/repo")
import bench
import acc_genomics_amd as A
from acc_genomics_amd import synth
reads, haps = bench.make_c1(0)
with A.Context(0) as ctx:
    with A.PhmmBatch(ctx, [(synth.serialize_reads(reads), synth.serialize_haps(haps))]) as b:
        for rep in range(3):
            for _ in range(5): b.run(0)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): b.run(0)
            ctx.synchronize()
            print("probe=%s: %.4f ms per step" % (os.environ.get("ACCG_PHMM_PROBE", "1"), (time.perf_counter() - t0) / 20 * 1e3))
        k, s = b.time_in_step(0, 20)
        print("time_in_step: kernel %.4f step %.4f" % (k, s))
        raw, l10, cnt = b.results()
        print("rescued", cnt.rescued)
