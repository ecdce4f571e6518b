#!/usr/bin/env python3
"""cProfile of a whole `lrbinner.py reads` run on the 432 k-read stand-in: where the clustering stage spends its time.
python scripts/cluster_profile.py [n_reads]"""
import cProfile, io, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synth_sim8, write_fasta

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 432_333
reads, origin = synth_sim8(scale=n_reads / 40350.0)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    fa = os.path.join(tmp, "reads.fasta")
    write_fasta(fa, reads)
    del reads
    out = os.path.join(tmp, "out")
    import lrbinner
    argv = ["reads", "-r", fa, "-o", out, "-k", "3", "-bc", "10", "-bs", "2", "--ae-dims", "4", "--ae-epochs", "200",
            "-bit", "0", "-mbs", "5000", "--cuda", "-t", "16"]
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    try:
        lrbinner.main(argv)
    except SystemExit:
        pass
    pr.disable()
    print(f"wall {time.time() - t0:.2f} s")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue()[:9000])
