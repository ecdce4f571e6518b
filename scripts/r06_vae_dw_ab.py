#!/usr/bin/env python3
"""A/B of the large-batch dW kernel of the fused VAE step (LRB_VAE_DW_FORM=0 round-5 form with 24 B/lane of scratch, 1 two
workgroups a CU, 2 BatchNorm table first): microseconds per optimiser step by batch size for the C1 and C3 networks
(bench.vae_step_times), each form in a child process, twice in turn.  -> gpurun_out/r06_vae_dw_ab.txt"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, json; sys.path.insert(0, %r); import torch, bench; from lrbinner_amd import device as lrb; "
        "print(json.dumps(bench.vae_step_times(torch, lrb)))" % ROOT)
rows = []
for rep in range(2):
    for form in ("0", "1", "2"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LRB_VAE_DW_FORM=form), capture_output=True, text=True, cwd=ROOT)
        if r.returncode != 0:
            rows.append(f"form {form}: FAILED {r.stderr[-400:]}")
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rows.append(f"rep {rep} form {form}: " + "  ".join(f"{sh[:2]} b{bs} {e['us']:.1f}" for sh in ("c1_shape", "c3_shape") for bs, e in d[sh].items()))
        print(rows[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r06_vae_dw_ab.txt"), "w").write(__doc__ + "\n" + "\n".join(rows) + "\n")
