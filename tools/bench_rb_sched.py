#!/usr/bin/env python3
"""receive_byte from host memory: how the call's sub-batch schedule (rxloop.hip: receive_byte_any; MERCURY_RB_SCHED=<a,b,c,..>) moves the rate,
per sample format. Windows from the library's own passband self-simulation at a clean point (what bench.py's receive_byte record uses).
  python tools/bench_rb_sched.py [cfg=8] [W=1024]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mercury_amd import RxPhy  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rb = RxPhy(cfg, max_batch=W)
    _, wins, _ = rb.passband_test_esn0([30.0], W, 1500.0, seed=0x4D455243, want_windows=True)
    w32 = np.rint(np.clip(wins, -1.0, 1.0) * 2147483647.0).astype(np.int32)
    w16 = np.rint(np.clip(wins, -1.0, 1.0) * 32767.0).astype(np.int16)

    def med(x, reps=4):
        rb.receive_byte(x, 1500.0); rb.receive_byte(x, 1500.0)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = rb.receive_byte(x, 1500.0)
            ts.append(time.perf_counter() - t0)
        return W / sorted(ts)[len(ts) // 2], r

    ref = {}
    res = {}
    for sched in ("default", "256", "512", "128,384,512", "128,256,640", "192,320,512", "64,192,768", "128,896", "256,768"):
        if sched == "default":
            os.environ.pop("MERCURY_RB_SCHED", None)
        else:
            os.environ["MERCURY_RB_SCHED"] = sched
        row = {}
        for name, x in (("f64", wins), ("int32", w32), ("int16", w16)):
            rate, r = med(x)
            key = r["payload"].tobytes() + r["stats"].tobytes()
            ref.setdefault(name, key)
            row[name] = round(rate)
            row[name + "_same"] = key == ref[name]
        res[sched] = row
        print(sched, row, file=sys.stderr)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
