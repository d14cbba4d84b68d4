#!/usr/bin/env python3
"""Regenerates tests/golden/balancer_*.txt from the reference's own balancer
(oracle/_ref/gen_balancer_script, linked against /root/reference/src/core compiled unmodified).
Run in the build container, where /root/reference exists:  python tests/golden/make_balancer_scripts.py"""
import os
import subprocess

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "ref"], stdout=subprocess.DEVNULL)
for algo in ("adaptive", "aimd", "fixed"):
    out = subprocess.check_output([os.path.join(root, "oracle", "_ref", "gen_balancer_script"), algo], text=True)
    with open(os.path.join(here, "balancer_%s.txt" % algo), "w") as f:
        f.write(out)
    print(algo, [int(l.split()[1]) // 1000 for l in out.splitlines()])
