#!/usr/bin/env python3
"""Build container only.  The REFERENCE's cluster search (mbcclr_utils.cluster_utils.cluster_points, imported from
/root/reference) on a latent.npy trained by this build on the GPU box, under random.seed(1..N): the sizes of the
clusters it finds, seed by seed -- next to what this build's search found on the same latents under the same seeds
(scripts/e2e_merge_probe.py writes both inputs).  Shows whether a run that merges two genomes does so because of
the latents or because of the search, and that the two searches agree seed by seed.

    python tests/golden/ref_recluster.py LATENT.npy OURS.txt [n_seeds] > tests/golden/e2e_recluster_big.json"""
import json
import os
import random
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
from make_golden_sim8 import import_reference  # noqa: E402

lat = np.load(sys.argv[1])
ours = {int(l.split(":")[0].split()[1]): [int(x) for x in l.split(":")[1].split()] for l in open(sys.argv[2]) if l.strip()}
n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
_, cu = import_reference()
rows = []
for s in range(1, n_seeds + 1):
    random.seed(s)
    clusters = cu.cluster_points(lat.copy(), 0, 5000)
    sizes = [len(v) for v in clusters.values()]
    big = lambda xs: [x for x in xs if x > 5000]
    rows.append({"seed": s, "reference_cluster_sizes": sizes, "this_build_cluster_sizes": ours.get(s),
                 "reference_bins": len(big(sizes)), "this_build_bins": len(big(ours.get(s, [])))})
    print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
print(json.dumps({"latents": "this build, 432 k-read stand-in (scripts/e2e_merge_probe.py), 4 latent dimensions",
                  "search": "cluster_points(latent, iterations=0, min_cluster_size=5000) under random.seed(seed)",
                  "runs": rows}, indent=1))
