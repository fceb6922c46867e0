#!/usr/bin/env python3
"""Transcribes SURVEY.md section 0's per-mode parameter table into tests/golden/survey_mode_table.json.

That table was printed by the surveyor from the COMPILED reference after the real cl_telecom_system::load_configuration(cfg)
(telecom_system.cc:2487-3025) — a translation unit oracle/_ref cannot link here (audio / GUI dependencies), which is why
oracle/ref_harness.cc restates the 17 mode rows. The transcription pins the mode table independently of
that restatement: tests assert that the oracle, the library's host-side table builder (mgpu_host_mode_info) and a live
context (mgpu_get_info) all report exactly these numbers. Run from the repo root: python tests/golden/make_survey_mode_table.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MOD = {"BPSK": 2, "QPSK": 4, "8PSK": 8, "16QAM": 16, "32QAM": 32}
rows = []
for line in open(os.path.join(ROOT, "SURVEY.md")):
    if re.match(r"^\| \**\d+\** \|", line):
        cells = [c.strip().strip("*") for c in line.strip().strip("|").split("|")]
        if len(cells) == 16:
            rows.append(cells)
assert len(rows) == 17, len(rows)
modes = {}
for r in rows:
    modes[str(int(r[0]))] = {"M": MOD[r[1].split()[0]], "K": int(r[2]), "P": int(r[3]), "Nsymb": int(r[4]), "nPilots": int(r[5]), "nData": int(r[6]),
                             "nBits": int(r[7]), "nVirtual": int(r[8]), "nReal": int(r[9]), "bit_blk": int(r[10]), "tf_blk": int(r[11]),
                             "preamble_nsymb": int(r[12]), "estimator": {"LS": 1, "ZF": 0}[r[13]], "amp_restore": {"yes": 1, "no": 0}[r[14]], "E": int(r[15])}
doc = {"provenance": "SURVEY.md section 0, 'Per-mode parameter table [probe - printed from the compiled reference after load_configuration(cfg)]': "
                     "the surveyor compiled /root/reference with the REAL cl_telecom_system::load_configuration (telecom_system.cc:2487-3025, which oracle/_ref "
                     "cannot link) and printed these members for CONFIG_0..16. Transcribed by tests/golden/make_survey_mode_table.py; independent of "
                     "oracle/ref_harness.cc's restatement of the mode rows. estimator: 1 = LS, 0 = ZERO_FORCE; E = Tanner-graph edges of the mode's LDPC code.",
       "fixed": {"N": 1600, "Nc": 50, "Nfft": 256, "Ngi": 16, "Nofdm": 272, "ls_window": 21},
       "modes": modes}
json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "survey_mode_table.json"), "w"), indent=1)
print("wrote %d modes" % len(modes))
