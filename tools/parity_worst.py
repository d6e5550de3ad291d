#!/usr/bin/env python
"""Condenses the JSON lines of tools/parity_report.py (stdin) into the worst figure per column."""
import json
import sys

rows = [json.loads(l) for l in sys.stdin if l.startswith('{"case"')]
keys = ["out_err_unexplained", "grad_rel_unexplained", "grad_row_rel_unexplained", "grad_l2", "flipped", "flipped_worst", "id_mismatch_unexplained"]
for group in ("s1", "fuzz", "s2", "s3", "s5"):
    rs = [r for r in rows if r["case"].startswith(group)]
    if not rs:
        continue
    print(group, len(rs), "cases:", " ".join("%s=%.3g (%s)" % (k, max(r[k] for r in rs), max(rs, key=lambda r: r[k])["case"].split(" P=")[0]) for k in keys))
