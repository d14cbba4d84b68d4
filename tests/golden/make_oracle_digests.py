#!/usr/bin/env python3
"""Writes tests/golden/oracle_stream_digests.json (sha256 of the oracle's access units for three
seeded clips).  Regenerate only when the encoder's non-normative decisions change on purpose."""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from oracle import oracle as O  # noqa: E402
from tests.test_oracle_roundtrip import run  # noqa: E402

cases = {"s2_176x144_qp28": (176, 144, 5, [28], "s2"), "s2_64x48_mixqp": (64, 48, 7, [30, 20, 40], "s2"), "s3_48x32_qp6": (48, 32, 4, [6], "s3")}
out = {k: run(O, w, h, n, q, kind=kind)[0] for k, (w, h, n, q, kind) in cases.items()}
json.dump(out, open(os.path.join(here, "oracle_stream_digests.json"), "w"), indent=1, sort_keys=True)
print(out)
