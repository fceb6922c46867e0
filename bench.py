#!/usr/bin/env python3
"""bench.py — RX frames/s + LDPC iterations/s for Mercury's physical-layer RX hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (one rank per GPU). A *step* is one pass of the whole hot path (OFDM demod -> estimate/equalise ->
  de-interleave -> demap -> LDPC -> de-scramble/CRC) over one batch of synthetic frames that is already
  resident in HBM. Frames are sharded by frame index across ranks (weak scaling: every rank owns
  --frames frames per step); there is no data-path collective, only the barrier + MAX-time reduction
  the contract asks for.

Workload (BASELINE.json configs[1]): 4096 mode-8 frames (QPSK, LDPC 6/16, N=1600), max 50 iterations,
AWGN. Default Es/N0 = -15 dB so that every frame executes all 50 iterations ("mode 8 @ 50 iters");
--esn0 2.5 gives the operating point with early termination. Decoder = sum-product in double (the
reference's algorithm, results identical to the CPU path); --decoder minsum selects the fp32 variant.

Output (rank 0): the LAST stdout line is the contract record - one JSON object of at most 4096 bytes (compact_line(): metric .. config,
kernel_ms, roofline, cpu_baseline, and the same workload's two other points: operating_point = threshold + 3 dB with early termination,
waterfall_point = just below the threshold, where every frame runs all the iterations on LLRs of real magnitude). The full record (other
decoders, opcode classes, per-rank detail, PCIe pipeline, receive_byte, machine) is written to bench_extras.json beside this script and
printed as an EARLIER stdout line (--line compact|full|both). `roofline` prices the dominant kernel (the LDPC decoder) with SURVEY.md
§8d's algorithmic bytes (16*E + 4*N per codeword-iteration + LLRs in + payload out) against 8 TB/s; EVERY `frac` of the record is that
model fraction (an effective bandwidth: the messages are LDS-resident, `traffic` is what the PMC counters saw); the bound the kernels
really hit, vector-instruction issue, is `roofline.secondary.frac`. `cpu_baseline` times the CPU checker (the plain-C port of the
reference, bit-identical to it) on a bounded sample of the very same frames on the host cores, and cross-checks the GPU results on them.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

from mercury_amd import DEC_GBF, DEC_MINSUM, DEC_SPA, DEC_SPA_FAST, RxPhy, RxPool, STATS_DTYPE

DECODERS = {"spa": DEC_SPA, "minsum": DEC_MINSUM, "gbf": DEC_GBF, "spa_fast": DEC_SPA_FAST}
from mercury_amd.sharding import frame_range

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md
WATERFALL_BELOW_THRESHOLD_DB = 1.5     # waterfall_point: Es/N0 = the mode's threshold (tests/conftest.py OPERATING_ESN0 - 2 dB) minus this
SEED = 0x4D455243


def algorithmic_bytes(rx, iters_exec_sum, frames):
    """B_frame = 16*Nsymb*Nofdm + 4*N + I_exec*(16*E + 4*N) + ceil(nReal/8) + 16."""
    b_in = 16 * rx.Nsymb * rx.Nofdm
    b_iter = 16 * rx.E + 4 * rx.N
    b_out = rx.payload_stride + 16
    ldpc = frames * (4 * rx.N + b_out) + iters_exec_sum * b_iter       # what the decoder kernel itself moves
    total = frames * (b_in + 4 * rx.N + b_out) + iters_exec_sum * b_iter
    return ldpc, total, b_iter


PROFILE_ROUND = "r06"          # profiles/<round>_instruction_mix.json (PMC passes) and <round>_valu_cycles.json (opcode issue costs)


def machine_of(device_index):
    """Compute units, SIMDs and engine clock of the device being timed, from the HIP runtime (mgpu_device_props_get), so that the issue-cycle
    budget below is this box's and not a constant; also what the pool's placement uses (PCI address, NUMA node)."""
    from mercury_amd import device_props
    p = device_props(device_index)
    return {"name": p["name"], "gcn_arch": p["gcn_arch"], "compute_units": p["compute_units"], "simds": 4 * p["compute_units"],
            "clock_hz": 1e3 * p["clock_khz"], "pci_bus_id": p["pci_bus_id"], "numa_node": p["numa_node"]}


def live_hbm_traffic(args):
    """HBM bytes per launch of the two kernels of THIS workload on THIS box: (2 x FETCH_SIZE + WRITE_SIZE) KB - FETCH_SIZE doubled as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads - from two rocprofv3 counter passes (one counter per
    pass, --kernel-trace only: the guide's recipe, never combined with other trace domains) of a two-step run of this same command. A regression
    in real HBM traffic then shows in the line of the run that has it, not only in a committed profile. Returns {kernel: bytes} or None
    (no rocprofv3, a pass failed or timed out: the committed profile's figure is quoted instead)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    # this process is itself being profiled (rocprofv3 -- python bench.py ...): no profiler inside a profiler
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    work = tempfile.mkdtemp(prefix="mgpu_pmc_", dir="/tmp")
    raw = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, ctr)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-live-pmc", "--cfg", str(args.cfg), "--frames", str(args.frames),
                   "--iters", str(args.iters), "--esn0", str(args.esn0), "--decoder", args.decoder, "--variant", args.variant, "--channel", str(args.channel)]
            if args.ldpc_only:
                cmd.append("--ldpc-only")
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=90, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return None
            acc = {}
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == ctr:
                        acc.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
            raw[ctr] = {k: sum(v) / len(v) for k, v in acc.items()}
        res = {}
        for k in raw["FETCH_SIZE"]:
            if k in raw["WRITE_SIZE"] and ("ldpc" in k or "frontend" in k):
                res[k] = (2.0 * raw["FETCH_SIZE"][k] + raw["WRITE_SIZE"][k]) * 1024.0
        return res or None
    except Exception:
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def profile_mix(key, workload_ok=True):
    """The committed PMC passes of one kernel launch (tools/collect_pmc_mix.sh; key "spa" / "spa_fast" / "minsum" = the decoder on the
    headline workload, "<decoder>_op" = on the operating-point workload, "frontend"), or None when they were taken from another build of
    the decoder kernels than the one being timed (stamp = mercury_amd.build.decoder_digest())."""
    name = "profiles/%s_instruction_mix.json" % PROFILE_ROUND
    if not workload_ok:
        return None, "no profile for this workload"
    try:
        from mercury_amd.build import decoder_digest
        prof = json.load(open(os.path.join(ROOT, name)))
        if prof.get("decoder_digest") != decoder_digest():
            return None, "%s was taken from another build of the decoder kernels (stamp %s, library %s)" % (name, prof.get("decoder_digest"), decoder_digest())
        if key not in prof:
            return None, "%s holds no PMC passes of this workload (%s)" % (name, key)
        return prof[key], "%s[%s] (PMC, same workload, same decoder build %s)" % (name, key, prof["decoder_digest"])
    except Exception as e:
        return None, "profile unreadable: %r" % (e,)


def issue_view(mix, src, kernel_ms, machine, sclk=None):
    """Secondary view for the bound these kernels actually hit (vector-instruction issue). The dynamic opcode mix of one launch comes from
    the committed PMC passes of THAT workload, the issue cost of each opcode class from the micro-benchmark measured on the same kind of
    box (profiles/<round>_valu_cycles.json, tools/ubench/valu_cycles.hip); issue cycles needed = sum(count x cost), available = the
    device's SIMDs x engine clock (HIP runtime) x kernel time of this run. Two readings: "isolated" prices every class at its back-to-back
    cost (32-bit operations 2 cycles), "slots" at one 4-cycle issue slot per instruction (what a mixed fp64 stream pays: MIX_FMA_CND in
    the micro-benchmark) - the truth lies between them."""
    cname = "profiles/%s_valu_cycles.json" % PROFILE_ROUND
    cfile = os.path.join(ROOT, cname)
    if mix is None or not os.path.exists(cfile):
        return {"unavailable": src if mix is None else cname + " missing"}
    try:
        cyc = {k.split("/")[0]: v["cycles_at_2p4ghz"] for k, v in json.load(open(cfile))["results"].items() if k.endswith("/8w")}
        fp64 = mix["SQ_INSTS_VALU_ADD_F64"] + mix["SQ_INSTS_VALU_MUL_F64"] + mix["SQ_INSTS_VALU_FMA_F64"]
        fp32 = mix["SQ_INSTS_VALU_ADD_F32"] + mix["SQ_INSTS_VALU_MUL_F32"] + mix["SQ_INSTS_VALU_FMA_F32"]
        classes = {"fp64": (fp64, cyc["FMA_F64"]), "trans_f64": (mix["SQ_INSTS_VALU_TRANS_F64"], cyc["RCP_F64"]),
                   "fp32": (fp32, cyc["FMA_F32"]), "trans_f32": (mix["SQ_INSTS_VALU_TRANS_F32"], cyc["RCP_F32"]),
                   "int32": (mix["SQ_INSTS_VALU_INT32"], cyc["AND_B32"]), "cvt": (mix["SQ_INSTS_VALU_CVT"], cyc["CVT_F64_I32"]),
                   "int64": (mix["SQ_INSTS_VALU_INT64"], cyc["LSHL_ADD"])}
        rest = mix["SQ_INSTS_VALU"] - sum(c for c, _ in classes.values())
        classes["compare_select_move"] = (rest, 0.5 * (cyc["CMP_U32"] + cyc["MOV_B32"]))
        isolated = sum(c * w for c, w in classes.values())
        slots = sum(c * max(w, cyc["FMA_F64"]) for c, w in classes.values())
        # the engine clock the device actually held while this kernel was timed (hwmon samples) when there are any; else the runtime's maximum
        clock_hz = 1e6 * sclk["median"] if sclk and sclk.get("median") else machine["clock_hz"]
        avail = machine["simds"] * clock_hz * kernel_ms * 1e-3
        # "frac" is the counters' own, clock-free reading where the profile has it (valu_busy_pmc below); the two priced readings bracket it and can
        # exceed 1 by the few per cent the sampled clock and the micro-benchmarked costs are off by
        busy = 4.0 * mix["SQ_ACTIVE_INST_VALU"] / (mix["GRBM_GUI_ACTIVE"] / 8.0 * machine["simds"]) if mix.get("GRBM_GUI_ACTIVE") and fp32 == 0 else None
        return {"bound": "valu_issue", "unit": "SIMD issue cycles per launch", "available": avail,
                "needed_isolated_costs": isolated, "frac_isolated": isolated / avail,
                "needed_4cycle_slots": slots, "frac_slots": slots / avail, "frac": busy if busy is not None else slots / avail,
                "clock_hz_used": clock_hz, "clock_source": "hwmon freq1_input, median over the timed region" if clock_hz != machine["clock_hz"] else "hipDeviceProp clockRate (maximum)",
                "valu_instructions_per_launch": mix["SQ_INSTS_VALU"],
                # PMC only, clock-free: SQ_ACTIVE_INST_VALU counts issue quanta; x4 turns them into SIMD cycles for fp64 streams (an over-estimate for 32-bit
                # operations, which issue in about 2.3 cycles: the fp32 decoders can read above 1), GRBM_GUI_ACTIVE / 8 XCDs are the launch's cycles
                "valu_busy_pmc": 4.0 * mix["SQ_ACTIVE_INST_VALU"] / (mix["GRBM_GUI_ACTIVE"] / 8.0 * machine["simds"]) if mix.get("GRBM_GUI_ACTIVE") else None,
                "machine": {k: machine[k] for k in ("compute_units", "simds", "clock_hz")},
                "classes": {k: {"count": c, "cycles_each": w} for k, (c, w) in classes.items()},
                "source": "opcode mix: %s; costs: %s (micro-benchmark); SIMDs and clock: hipDeviceProp; time: this run" % (src, cname)}
    except Exception as e:                       # a stale profile must not break the bench line
        return {"error": repr(e)}


class ClockSampler(threading.Thread):
    """Engine clock (and board power) of the device while the timed region runs, from the amdgpu hwmon files of its PCI address: a clock
    drop (power cap, thermal) then shows in the bench line instead of silently moving the roofline fraction."""

    def __init__(self, pci_bus_id, period=0.01):
        super().__init__(daemon=True)
        import glob
        base = "/sys/bus/pci/devices/%s" % pci_bus_id.lower()
        self.freq = (glob.glob(base + "/hwmon/hwmon*/freq1_input") or [None])[0]
        self.power = (glob.glob(base + "/hwmon/hwmon*/power1_average") + glob.glob(base + "/hwmon/hwmon*/power1_input") or [None])[0]
        self.period, self.mhz, self.watt, self.stop_flag = period, [], [], threading.Event()

    def run(self):
        while not self.stop_flag.is_set():
            try:
                if self.freq:
                    self.mhz.append(int(open(self.freq).read()) / 1e6)
                if self.power:
                    self.watt.append(int(open(self.power).read()) / 1e6)
            except Exception:
                pass
            self.stop_flag.wait(self.period)

    def summary(self):
        self.stop_flag.set()
        self.join(timeout=1.0)
        if not self.mhz:
            return None
        m = sorted(self.mhz)
        out = {"min": m[0], "median": m[len(m) // 2], "max": m[-1], "samples": len(m), "source": "hwmon freq1_input"}
        if self.watt:
            out["power_w_median"] = sorted(self.watt)[len(self.watt) // 2]
        return out


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


COMPACT_LIMIT = 4096          # bytes: the contract line is the LAST stdout line and must survive a driver that keeps a stdout tail


def _sig(x, n=6):
    """Numbers of the compact line carry n significant digits (ints, bools, None and strings pass through)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (n, float(x)))
    except (TypeError, ValueError):
        return x


def _pick(d, keys, n=6):
    return {k: _sig(d[k], n) for k in keys if isinstance(d, dict) and k in d}


def _roofline_compact(r):
    """bound / achieved / peak / unit / frac / traffic / kernel of one kernel's roofline; `frac` is ALWAYS the SURVEY.md §8d HBM-model fraction
    (algorithmic bytes / kernel time / 8 TB/s), the issue-bound reading lives in secondary.frac only."""
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "bytes_per_codeword_iteration", "bytes_per_frame"))
    sec = r.get("secondary")
    if isinstance(sec, dict) and "frac" in sec:
        out["secondary"] = _pick(sec, ("bound", "frac"), 4)
    return out


def _point_compact(p):
    """One secondary operating record (operating_point / waterfall_point) of the compact line."""
    if not isinstance(p, dict):
        return None
    out = _pick(p, ("value", "unit", "ms_per_step", "esn0_db", "avg_iters_per_frame", "frames_running_all_iters", "decoded_fraction", "ldpc_iters_per_s"))
    if "kernel_ms" in p:
        out["kernel_ms"] = _pick(p["kernel_ms"], ("frontend", "ldpc"), 5)
    rf = p.get("roofline") or {}
    if "decoder" in rf:
        out["roofline"] = _roofline_compact(rf["decoder"])
        fe = _roofline_compact(rf.get("frontend"))
        if fe:
            out["roofline_frontend"] = _pick(fe, ("frac", "traffic", "secondary"))
    return out


def compact_line(full):
    """The contract record: what the driver parses (metric .. config, roofline, cpu_baseline) plus the two other first-class points of the same
    workload (operating_point: threshold + 1 dB with early termination; waterfall_point: the Es/N0 at which >= 90 % of the frames run all the
    iterations on LLRs of real magnitude), at most COMPACT_LIMIT bytes whatever the full record holds. Everything else (other decoders, opcode
    classes, per-rank detail, PCIe pipeline, machine) stays in the full record: bench_extras.json beside this script and an EARLIER stdout line."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: _sig(full[k], 7) for k in keep if k in full}
    cfg = full.get("config", {})
    out["config"] = {k: cfg[k] for k in ("workload", "frames_per_step_per_gpu", "cfg", "esn0_db", "decoder", "parallelism") if k in cfg}
    out.update(_pick(full, ("ldpc_iters_per_s", "avg_iters_per_frame", "decoded_fraction", "hard_frames_per_step")))
    if "kernel_ms" in full:
        out["kernel_ms"] = _pick(full["kernel_ms"], ("frontend", "ldpc", "launches_averaged"), 5)
    out["roofline"] = _roofline_compact(full.get("roofline"))
    if isinstance(out["roofline"], dict) and isinstance(full.get("roofline"), dict):
        out["roofline"]["note"] = "sec. 8d model bytes; messages are LDS-resident (traffic = PMC HBM bytes); bound is vector issue: secondary"
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "ldpc_iters_per_s", "gpu_vs_cpu_mismatches", "reference_1core_frames_per_s", "gpu_over_cpu"))
    for name in ("operating_point", "waterfall_point"):
        pt = _point_compact(full.get(name))
        if pt:
            out[name] = pt
    ex = full.get("extras_per_gpu") or {}
    rb = ex.get("receive_byte_capture_windows")
    if isinstance(rb, dict) and "error" not in rb:
        out["receive_byte"] = _pick(rb, ("windows", "decoded", "windows_per_s_device_resident", "windows_per_s_host_buffers", "windows_per_s_host_buffers_int32_samples"), 5)
    for k in ("pcie_inclusive_frames_per_s", "pcie_inclusive_pinned_frames_per_s"):
        if k in full:
            out[k] = _sig(full[k], 5)
    pd = full.get("per_device") or full.get("per_rank")
    if isinstance(pd, list) and len(pd) > 1:          # N > 1: the clock every board held, its power and its own kernel / wall times
        out["per_device"] = [[_sig(r.get(k), 4) for k in ("sclk_mhz_median", "sclk_mhz_min", "power_w_median", "ldpc_kernel_ms", "frontend_kernel_ms", "wall_ms")] for r in pd]
        out["per_device_columns"] = "sclk_mhz_median, sclk_mhz_min, power_w_median, ldpc_kernel_ms, frontend_kernel_ms, wall_ms"
    clk = full.get("sclk_mhz_during_run")
    if isinstance(clk, dict):
        out["sclk_mhz"] = _pick(clk, ("min", "median", "max", "power_w_median"), 4)
    out["full_record"] = "bench_extras.json + the stdout line before this one"
    # a guard, not a plan: should the record ever outgrow the limit, drop the least important blocks until it fits
    for victim in ("per_device", "receive_byte", "pcie_inclusive_pinned_frames_per_s", "sclk_mhz", "waterfall_point", "operating_point"):
        if len(json.dumps(out)) <= COMPACT_LIMIT:
            break
        out.pop(victim, None)
        out.pop(victim + "_columns", None)
    return out


def emit(full, args):
    """Full record -> bench_extras.json beside the script (best effort) and, unless --line compact, an earlier stdout line; the compact contract
    record is the LAST stdout line."""
    compact = compact_line(full)
    assert len(json.dumps(compact)) <= COMPACT_LIMIT
    try:
        with open(os.path.join(ROOT, "bench_extras.json"), "w") as fh:
            json.dump(full, fh)
    except OSError:
        pass
    if args.line in ("full", "both"):
        print(json.dumps(dict({"bench_full_record": True}, **full)), flush=True)
    if args.line in ("compact", "both"):
        print(json.dumps(compact), flush=True)


def cpu_baseline(cfg, max_iters, bb_sample, flags, gpu_payload, gpu_stats):
    """Time the CPU checker on a bounded sample of the same frames; also cross-check the GPU output."""
    import oraclelib
    cores = usable_cores()
    n = bb_sample.shape[0]
    per = max(1, n // cores)
    used = min(cores, n)
    chunks = [(i * per, min(n, (i + 1) * per)) for i in range(used)]
    chunks = [c for c in chunks if c[1] > c[0]]
    ctxs = [oraclelib.Oracle(cfg, max_iters) for _ in chunks]
    results = [None] * len(chunks)

    def work(i):
        a, b = chunks[i]
        results[i] = ctxs[i].rx_many(bb_sample[a:b], flags)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(chunks))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    frames = sum(b - a for a, b in chunks)
    iters_exec = sum(r[0] for r in results)
    mism = 0
    for (a, b), r in zip(chunks, results):
        _, iters, crc, pl = r
        mism += int((gpu_stats[a:b, 0] != iters).sum()) + int((gpu_stats[a:b, 1] != crc).sum())
        mism += int((gpu_payload[a:b, : pl.shape[1]] != pl).any(axis=1).sum())
    out = {"value": frames / dt, "unit": "frames/s", "cores": len(chunks), "kind": "port",
           "ldpc_iters_per_s": iters_exec / dt,
           "sample": "%d of the benchmarked frames (same samples, same flags), %d threads x %d frames, %.1f s wall"
                     % (frames, len(chunks), per, dt),
           "gpu_vs_cpu_mismatches": mism}
    # the real reference objects (oracle/_ref), one thread, for the record
    if oraclelib.RefLib.available():
        try:
            ref = oraclelib.RefLib(cfg, max_iters)
            k = min(8, n)
            t0 = time.perf_counter()
            for f in range(k):
                ref.rx(bb_sample[f], flags)
            out["reference_1core_frames_per_s"] = k / (time.perf_counter() - t0)
        except Exception as e:  # pragma: no cover
            out["reference_1core_error"] = str(e)
    return out


def headline_workload(args, F):
    return (not args.ldpc_only) and args.cfg == 8 and F == 4096 and args.iters == 50 and args.variant == "receive_byte" and args.channel == 0


def operating_point_roofline(rx, decoder, m, F, machine, profiled, sclk=None, mix_suffix="_op"):
    """Rooflines of the two kernels of one operating-point step (SURVEY.md §8d C2's second point): the front-end is a streaming kernel
    - its input samples against HBM - and also priced against vector issue from its PMC mix; the decoder (LDS-resident messages) against
    vector issue from a PMC pass of THIS launch (profiles/<round>_instruction_mix.json["<decoder>_op"]), with the §8d HBM-model figure
    beside it for comparison with the headline."""
    b_in = 16 * rx.Nsymb * rx.Nofdm
    fe = {"kernel": "mgpu_frontend_kernel", "bound": "hbm", "achieved": F * b_in / (m["frontend_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
          "bytes_per_frame": b_in}
    fe["frac"] = fe["achieved"] / fe["peak"]
    mix, src = profile_mix("frontend", profiled)
    fe["traffic"] = (2.0 * mix["FETCH_SIZE"] + mix["WRITE_SIZE"]) * 1024.0 if mix else None
    fe["secondary"] = issue_view(mix, src, m["frontend_ms"], machine, sclk)
    ldpc_bytes, _, b_iter = algorithmic_bytes(rx, m["avg_iters"] * F, F)
    mix, src = profile_mix(decoder + mix_suffix, profiled)
    # `frac` is the §8d HBM-model fraction here as everywhere in the line (an EFFECTIVE bandwidth: the messages are LDS-resident); the bound these
    # launches actually hit - vector-instruction issue - is secondary.frac (the counters' VALU-busy reading where the profile has this launch)
    ach = ldpc_bytes / (m["ldpc_ms"] * 1e-3)
    dec = {"kernel": "mgpu_ldpc_%s_kernel" % decoder, "bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
           "bytes_per_codeword_iteration": b_iter, "traffic": (2.0 * mix["FETCH_SIZE"] + mix["WRITE_SIZE"]) * 1024.0 if mix else None, "traffic_source": src,
           "secondary": issue_view(mix, src, m["ldpc_ms"], machine, sclk)}
    return {"frontend": fe, "decoder": dec}


def extras(args, rx, bufs, payload, stats, stream, dev, F, noise_amp, machine):
    """Secondary measurements on rank 0's GPU, outside the contract's timed region: the other decoders on the same
    inputs, and every decoder at the mode's operating point (threshold + 3 dB) where early termination works."""
    from conftest import OPERATING_ESN0
    out = {}

    def timed(phy, inputs, steps=10):
        for i in range(2):
            phy.receive_dev(inputs[i % len(inputs)].data_ptr(), F, payload.data_ptr(), stats.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        phy.enable_timing(True)
        sampler = ClockSampler(machine["pci_bus_id"], period=0.002)
        sampler.start()
        hard0 = phy.decoder_hard_frames()
        t0 = time.perf_counter()
        for i in range(steps):
            phy.receive_dev(inputs[i % len(inputs)].data_ptr(), F, payload.data_ptr(), stats.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sclk = sampler.summary()
        fe, dec, _ = phy.kernel_ms_avg()
        phy.enable_timing(False)
        hard = (phy.decoder_hard_frames() - hard0) / steps          # frames per step decided without iterating (reported: max + 1 iterations)
        it = (float(stats[:, 0].clamp(max=args.iters).sum().item()) - hard * args.iters) / F      # iterations EXECUTED per frame
        ok = float(stats[:, 3].sum().item()) / F
        return {"frames_per_s": F * steps / dt, "ms_per_step": dt / steps * 1e3, "frontend_ms": fe, "ldpc_ms": dec, "avg_iters": it, "decoded_fraction": ok,
                "hard_frames_per_step": hard, "sclk_mhz": sclk}

    agc, vs = (1, 1) if args.variant == "receive_byte" else (0, 0)
    profiled = headline_workload(args, F)
    others = {}
    for other in [d for d in ("spa", "spa_fast", "minsum") if d != args.decoder]:
        rx2 = RxPhy(args.cfg, max_iters=args.iters, decoder=DECODERS[other], agc=agc, variance_source=vs, device=dev.index, max_batch=F)
        m = timed(rx2, bufs)
        # these kernels keep their messages in LDS: pricing them against HBM says nothing (the model fraction exceeds 1 on some
        # modes). What bounds them is vector-instruction issue: the fraction of the SIMDs' issue cycles their PMC opcode mix needs.
        mix, src = profile_mix(other, profiled and abs(m["avg_iters"] - 50.0) < 1e-6)
        sec = issue_view(mix, src, m["ldpc_ms"], machine, m["sclk_mhz"])
        lb, _, _ = algorithmic_bytes(rx, m["avg_iters"] * F, F)
        m["roofline_frac"] = lb / (m["ldpc_ms"] * 1e-3) / HBM_PEAK       # §8d model, like every `frac` of the line (not a ceiling for LDS-resident fp32 messages: can exceed 1)
        if sec and "frac" in sec:
            m["roofline_secondary"] = {"bound": "valu_issue", "frac": sec["frac_isolated"], "source": "isolated opcode costs; %s" % sec["source"]}
        out["same_inputs_decoder_" + other] = m
        others[other] = rx2
    op = OPERATING_ESN0[args.cfg] + 1.0
    amp = float(10.0 ** (-op / 20.0) / np.sqrt(2.0))
    bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
    rx.txgen_dev(SEED, 1 << 40, F, amp, bb.data_ptr(), None, channel=args.channel, stream=stream)
    torch.cuda.synchronize()
    for name, phy in [(args.decoder, rx)] + list(others.items()):
        r = timed(phy, [bb], steps=20)
        r["esn0_db"] = op
        r["roofline"] = operating_point_roofline(rx, name, r, F, machine, profiled, r["sclk_mhz"])
        out["operating_point_decoder_" + name] = r
    # the waterfall: the Es/N0 just below the mode's threshold, where (nearly) every frame runs ALL the iterations on LLRs of real magnitude (the
    # lanes spread over all of fdlibm's cases, unlike the noise-only headline input) - where the reference spends its "~35 ms per trial"
    wf = args.waterfall_esn0 if args.waterfall_esn0 is not None else OPERATING_ESN0[args.cfg] - 2.0 - WATERFALL_BELOW_THRESHOLD_DB
    bw = torch.empty_like(bb)
    rx.txgen_dev(SEED, 1 << 41, F, float(10.0 ** (-wf / 20.0) / np.sqrt(2.0)), bw.data_ptr(), None, channel=args.channel, stream=stream)
    torch.cuda.synchronize()
    r = timed(rx, [bw], steps=10)
    r["esn0_db"] = wf
    r["frames_running_all_iters"] = float((stats[:, 0] >= args.iters).sum().item()) / F
    r["roofline"] = operating_point_roofline(rx, args.decoder, r, F, machine, profiled and r["frames_running_all_iters"] > 0.9, r["sclk_mhz"], mix_suffix="_wf")
    out["waterfall_point_decoder_" + args.decoder] = r
    del bw
    # one frame per call through the blocking host-buffer entry point (mgpu_rx_batch, F = 1): what a receive_byte
    # patched as in INTEGRATION.md §1.2 waits for, PCIe copies and launch overheads included
    for name, src in (("worst_case", bufs[0]), ("operating_point", bb)):
        one = src[:1].cpu().numpy().view(np.complex128).reshape(1, -1)
        lat = []
        for i in range(12):
            t0 = time.perf_counter()
            rx.receive(one)
            lat.append((time.perf_counter() - t0) * 1e3)
        out["single_frame_latency_ms_" + name] = float(np.median(lat[2:]))
    for phy in others.values():
        phy.close()
    # the caller's side of the path: the whole receive_byte — mixer / filter, time and frequency synchronisation,
    # gates and retries, the RX path above — on capture windows of passband audio that the library's own passband self-simulation
    # (passband_test_EsN0: transmit_byte -> AWGN with delay) produced at a clean point; windows resident in HBM, and in host memory
    try:
        W = 1024 if args.cfg < 100 else 256
        rb = RxPhy(args.cfg, max_iters=args.iters, decoder=DECODERS[args.decoder], device=dev.index, max_batch=W)
        _, wins, _ = rb.passband_test_esn0([30.0], W, 1500.0, seed=SEED, want_windows=True)
        dwin = torch.from_numpy(wins).to(dev)
        torch.cuda.synchronize()

        def med(f, reps=3):
            f()
            f()                                                       # staging buffers and workspaces grow on the first calls
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = f()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2], r
        t_dev, r = med(lambda: rb.receive_byte_dev(dwin.data_ptr(), W, 1500.0))
        t_host, _ = med(lambda: rb.receive_byte(wins, 1500.0))
        # the audio device's own samples (the reference captures INT32, audioio.c:744, and widens on the host): half the bytes over PCIe
        w32 = np.rint(np.clip(wins, -1.0, 1.0) * 2147483647.0).astype(np.int32)
        t_i32, r32 = med(lambda: rb.receive_byte(w32, 1500.0))
        out["receive_byte_capture_windows"] = {"windows": W, "samples_per_window": int(wins.shape[1]), "decoded": int(r["stats"]["message_decoded"].sum()),
                                               "windows_per_s_device_resident": W / t_dev, "windows_per_s_host_buffers": W / t_host,
                                               "windows_per_s_host_buffers_int32_samples": W / t_i32,
                                               "decoded_int32_samples": int(r32["stats"]["message_decoded"].sum())}
        rb.close()
    except Exception as e:                                          # a secondary measurement must never cost the bench line
        out["receive_byte_capture_windows"] = {"error": str(e)[:200]}
    return out


def run_pool(args):
    """The same measurement with the host side the reference would have: ONE process (the C / C++ caller's shape,
    RX_SHM_process_main telecom_system.cc:2266-2390) driving N devices through include/mercury_pool.h - a context and a host worker
    thread per device, every device's shard resident in its own memory, no collectives. Weak scaling like the torchrun form: every
    device owns --frames frames per step. A step is one blocking mgpu_pool_rx_batch_dev call (all devices, including the read-back
    of the 24-byte stats records the merged counters are made from)."""
    N, F = args.gpus, args.frames
    devices = [0] * N if args.share_device else list(range(N))
    decoder = DECODERS[args.decoder]
    agc, vs = (1, 1) if args.variant == "receive_byte" else (0, 0)
    pool = RxPool(args.cfg, devices, max_iters=args.iters, decoder=decoder, agc=agc, variance_source=vs, max_batch=F)
    one = RxPhy(args.cfg, max_iters=args.iters, decoder=decoder, agc=agc, variance_source=vs, device=devices[0], max_batch=1)   # mode constants
    counts = [F] * N
    noise_amp = float(10.0 ** (-args.esn0 / 20.0) / np.sqrt(2.0))
    nbuf = max(1, args.nbuf)
    fs, ps = pool.frame_samples, pool.payload_stride
    d_in = []
    for b in range(nbuf):
        if args.ldpc_only:
            bufs = []
            for g in range(N):
                gen = torch.Generator(device="cuda:%d" % devices[g])
                gen.manual_seed(SEED + (b * N + g) * F)
                bufs.append(2.0 * torch.randn((F, pool.N), generator=gen, dtype=torch.float32, device="cuda:%d" % devices[g]))
            d_in.append(bufs)
        else:
            ptrs = [pool.device_malloc(g, F * fs * 16) for g in range(N)]
            pool.txgen_dev(SEED, b * F * N, counts, noise_amp, ptrs, None, channel=args.channel)
            d_in.append(ptrs)
    for g in range(N):
        torch.cuda.synchronize(devices[g])
    d_pay = [pool.device_malloc(g, F * ps) for g in range(N)]
    d_st = [pool.device_malloc(g, F * 24) for g in range(N)]
    d_it = [pool.device_malloc(g, F * 4) for g in range(N)]

    def step(i):
        src = d_in[i % nbuf]
        if args.ldpc_only:
            pool.ldpc_decode_dev([t.data_ptr() for t in src], counts, None, d_it)
        else:
            pool.receive_dev(src, counts, d_pay, d_st)
        return pool.counters()

    for i in range(max(args.warmup, 0)):
        step(i)
    pool.enable_timing(True)
    hard0 = pool.decoder_hard_frames()
    machines = [machine_of(d) for d in devices]
    samplers = [ClockSampler(m["pci_bus_id"]) for m in machines] if not args.share_device else [ClockSampler(machines[0]["pci_bus_id"])]
    for sm in samplers:              # one watcher per board: a clock sag under N boards' power must show next to roofline.frac
        sm.start()
    iters_total = decoded_total = 0
    dev_ms = np.zeros(N)
    for g in range(N):
        torch.cuda.synchronize(devices[g])
    t0 = time.perf_counter()
    for i in range(args.steps):
        k = step(i)                               # blocking: returns when every device has finished its shard
        iters_total += k["ldpc_iterations"]
        decoded_total += k["decoded"]
        dev_ms += np.array(k["device_ms"])
    for g in range(N):
        torch.cuda.synchronize(devices[g])
    dt = time.perf_counter() - t0
    clocks = [sm.summary() for sm in samplers]
    fe_ms, dec_ms, nl = pool.kernel_ms(0)
    per_device = []
    for g in range(N):
        fe_g, dec_g, _ = pool.kernel_ms(g)
        ck = clocks[0 if args.share_device else g] or {}
        per_device.append({"device": devices[g], "sclk_mhz_median": ck.get("median", -1.0), "sclk_mhz_min": ck.get("min", -1.0), "power_w_median": ck.get("power_w_median", -1.0),
                           "ldpc_kernel_ms": dec_g, "frontend_kernel_ms": fe_g, "wall_ms": float(dev_ms[g])})
    pool.enable_timing(False)
    hard_total = pool.decoder_hard_frames() - hard0        # frames decided without iterating: reported with max + 1 iterations, none executed
    iters_total -= hard_total * args.iters
    frames_total = F * N * args.steps
    iters_per_launch = iters_total / (args.steps * N)
    ldpc_bytes, _, b_iter = algorithmic_bytes(one, iters_per_launch, F)
    achieved = ldpc_bytes / (dec_ms * 1e-3)
    machine = machines[0]
    mix, mix_src = profile_mix(args.decoder, headline_workload(args, F) and abs(iters_per_launch - 50.0 * F) <= 1e-6 * F)
    line = {
        "metric": ("LDPC codewords/s (rate %d/1600, max %d iters)" % (pool.K, args.iters)) if args.ldpc_only else
                  ("RX frames/s (mode %d, max %d LDPC iters)" % (args.cfg, args.iters)),
        "value": frames_total / dt, "unit": "frames/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if args.decoder == "spa" else "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[4]-style decoder-only soak: %d rate-%d/1600 codewords per GPU per step, noise-only LLRs, "
                                "decoder=%s, max %d iterations" % (F, pool.K, args.decoder, args.iters)) if args.ldpc_only else
                               ("BASELINE.json configs[1]: %d mode-%d frames per GPU per step through %s at Es/N0 %+.1f dB, "
                                "%s variant, decoder=%s, max %d iterations" % (F, args.cfg, "2-path+AWGN" if args.channel else "AWGN",
                                                                               args.esn0, args.variant, args.decoder, args.iters)),
                   "frames_per_step_per_gpu": F, "cfg": args.cfg, "esn0_db": args.esn0, "decoder": args.decoder,
                   "parallelism": "pool x%d: one process, one context + host thread per device (include/mercury_pool.h), device-resident "
                                  "shards, no collectives%s" % (N, " [all contexts on GPU 0: --share-device]" if args.share_device else "")},
        "ldpc_iters_per_s": iters_total / dt, "avg_iters_per_frame": iters_total / frames_total,
        "decoded_fraction": decoded_total / frames_total,
        "hard_frames_per_step": hard_total / (args.steps * N),
        "kernel_ms": {"frontend": fe_ms, "ldpc": dec_ms, "launches_averaged": nl, "device": 0},
        "pool_device_ms_per_step": [float(x) / args.steps for x in dev_ms],
        "per_device": per_device, "sclk_mhz_during_run": clocks[0], "machine": machine,
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                     "traffic": (2.0 * mix["FETCH_SIZE"] + mix["WRITE_SIZE"]) * 1024.0 if mix else None, "traffic_source": mix_src,
                     "kernel": "mgpu_ldpc_%s_kernel" % args.decoder,
                     "secondary": issue_view(mix, mix_src, dec_ms, machine, clocks[0]),
                     "bytes_per_codeword_iteration": b_iter,
                     "note": "device 0's decoder launch; algorithmic bytes (SURVEY.md 8d: 16E+4N per codeword-iteration); messages are "
                             "LDS-resident so real HBM traffic is far lower; the decoders are bound by vector-instruction issue"},
    }
    if N == 1 and not args.no_cpu_baseline and not args.ldpc_only:
        import oraclelib
        cores = usable_cores()
        S = min(F, args.cpu_sample_per_core * cores)
        last = (args.steps - 1) % nbuf
        bb_h = pool.copy_to_host(0, np.zeros((S, fs), np.complex128), d_in[last][0])
        pay = pool.copy_to_host(0, np.zeros((S, ps), np.uint8), d_pay[0])
        st = pool.copy_to_host(0, np.zeros(S, STATS_DTYPE), d_st[0])
        st6 = np.stack([st["iterations_done"], st["crc"], st["all_zeros"], st["message_decoded"]], axis=1)
        flags = oraclelib.FLAGS_RECEIVE_BYTE if args.variant == "receive_byte" else oraclelib.FLAGS_BASEBAND_TEST
        line["cpu_baseline"] = cpu_baseline(args.cfg, args.iters, bb_h, flags, pay, st6)
        line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    line["placement"] = {"devices": devices, "numa_nodes": pool.numa_nodes(), "worker_threads": "bound to their device's NUMA node (MERCURY_POOL_AFFINITY=0 to disable)"}
    emit(line, args)
    pool.close()
    one.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cfg", type=int, default=8)
    ap.add_argument("--frames", type=int, default=4096, help="frames per step per GPU")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--esn0", type=float, default=-15.0)
    ap.add_argument("--decoder", choices=["spa", "minsum", "gbf", "spa_fast"], default="spa")
    ap.add_argument("--variant", choices=["receive_byte", "baseband_test"], default="receive_byte")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-per-core", type=int, default=160)
    ap.add_argument("--nbuf", type=int, default=2, help="distinct input batches cycled through")
    ap.add_argument("--channel", type=int, default=0, help="0 = AWGN, 1 = static 2-path + AWGN (BASELINE.json configs[3])")
    ap.add_argument("--ldpc-only", action="store_true",
                    help="BASELINE.json configs[4]: decoder-only soak on noise-only LLRs (every codeword runs --iters iterations)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary per-GPU measurements (min-sum, operating point)")
    ap.add_argument("--live-pmc", action="store_true",
                    help="N = 1 only: take two rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this workload (child processes, ~5 s) and report them as "
                         "roofline.traffic_live beside roofline.traffic (always the committed, stamped profile's figure)")
    ap.add_argument("--no-live-pmc", action="store_true", help="accepted for older command lines; the live passes are off unless --live-pmc")
    ap.add_argument("--waterfall-esn0", type=float, default=None,
                    help="Es/N0 of the waterfall_point record (default: the mode's threshold - %.1f dB)" % WATERFALL_BELOW_THRESHOLD_DB)
    ap.add_argument("--line", choices=["compact", "full", "both"], default="both",
                    help="stdout: 'both' (default) = the full record as one line, then the compact contract record (<= 4 KB) as the LAST line; "
                         "'compact' / 'full' = only that one")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for tests)")
    ap.add_argument("--share-device", action="store_true", help="testing only: every rank uses GPU 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing only: create the process group (and run the barrier / MAX / SUM reductions) even with one rank - the RCCL code path of the "
                         "driver's 8-GPU run on a one-GPU box")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the shard map of `--gpus N` (rank -> device, NUMA node when known, global frame ranges per input batch) as one JSON line and "
                         "exit; touches no GPU")
    ap.add_argument("--pool", action="store_true",
                    help="one process drives the --gpus devices through the C-ABI pool (include/mercury_pool.h: one context + one host "
                         "thread per device), device-resident shards; also what `python bench.py --gpus N` does when not under torchrun")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)
    if args.pool or (int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus > 1):
        return run_pool(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        local_rank = 0
    elif torch.cuda.device_count() > 0:
        local_rank %= torch.cuda.device_count()       # e.g. a launcher that exposes one device per rank
    collective = world > 1 or args.force_dist
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the reduction tensors live

    decoder = DECODERS[args.decoder]
    agc, vs = (1, 1) if args.variant == "receive_byte" else (0, 0)
    F = args.frames
    rx = RxPhy(args.cfg, max_iters=args.iters, decoder=decoder, agc=agc, variance_source=vs,
               device=local_rank, max_batch=F)
    noise_amp = float(10.0 ** (-args.esn0 / 20.0) / np.sqrt(2.0))
    stream = torch.cuda.current_stream().cuda_stream

    # synthetic inputs, born in HBM; frame indices are global so ranks hold disjoint frames
    nbuf = max(1, args.nbuf)
    bufs, sent = [], []
    for b in range(nbuf):
        lo, _ = frame_range(rank, world, F * world)
        frame0 = b * F * world + lo
        if args.ldpc_only:
            # noise-only LLRs (no codeword underneath): llr = 2y/sigma^2 with y ~ N(0, sigma^2), sigma = 1
            g = torch.Generator(device=dev)
            g.manual_seed(SEED + frame0)
            bufs.append(2.0 * torch.randn((F, rx.N), generator=g, dtype=torch.float32, device=dev))
            continue
        bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
        pl = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
        rx.txgen_dev(SEED, frame0, F, noise_amp, bb.data_ptr(), pl.data_ptr(), channel=args.channel, stream=stream)
        bufs.append(bb)
        sent.append(pl)
    payload = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
    stats = torch.empty((F, 6), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    iters_acc = torch.zeros((), dtype=torch.int64, device=dev)
    decoded_acc = torch.zeros((), dtype=torch.int64, device=dev)

    def step(i):
        if args.ldpc_only:
            rx.ldpc_decode_dev(bufs[i % nbuf].data_ptr(), F, d_payload=payload.data_ptr(), d_stats=stats.data_ptr(), stream=stream)
        else:
            rx.receive_dev(bufs[i % nbuf].data_ptr(), F, payload.data_ptr(), stats.data_ptr(), stream=stream)
        # iterations actually executed (max+1 means "never converged" after max iterations)
        iters_acc.add_(stats[:, 0].clamp(max=args.iters).sum())
        decoded_acc.add_(stats[:, 3].sum())

    for i in range(max(args.warmup, 0)):
        step(i)
    if args.warmup == 0:  # torch loads its reduction kernels lazily; keep that one-off out of the timed region
        (stats[:, 0].clamp(max=args.iters).sum() + stats[:, 3].sum()).item()
    torch.cuda.synchronize()
    if collective:
        dist.barrier()
    machine = machine_of(local_rank)
    sampler = ClockSampler(machine["pci_bus_id"])          # every rank watches its own board: a clock sag under N boards' power must be visible
    if sampler:
        sampler.start()
    rx.enable_timing(True)
    iters_acc.zero_()
    decoded_acc.zero_()
    hard0 = rx.decoder_hard_frames()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if collective:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sclk = sampler.summary() if sampler else None
    fe_ms, dec_ms, nl = rx.kernel_ms_avg()
    rx.enable_timing(False)
    # HARD frames (every |LLR| >= 200 and an odd parity check: the fp64 decoder decides them without iterating and reports max + 1 iterations,
    # as the reference's loop would after changing nothing) count with the iterations they EXECUTED: none
    hard_frames = rx.decoder_hard_frames() - hard0
    iters_acc.sub_(hard_frames * args.iters)

    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    sums = torch.stack([iters_acc, decoded_acc]).to(torch.float64).to(rdev)
    # per rank, next to roofline.frac in the N > 1 line: the engine clock its board held and the power it drew while timed, its own
    # kernel times and wall time (-1 = no hwmon file for that board)
    mine = torch.tensor([(sclk or {}).get("median", -1.0), (sclk or {}).get("min", -1.0), (sclk or {}).get("power_w_median", -1.0), dec_ms, fe_ms, dt * 1e3],
                        dtype=torch.float64, device=rdev)
    per_rank = [mine]
    if collective:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    dt = float(tmax.item())
    iters_total = float(sums[0].item())
    decoded_total = float(sums[1].item())
    frames_total = F * args.steps * world

    if rank == 0:
        iters_per_launch = float(iters_acc.item()) / args.steps
        ldpc_bytes, _, b_iter = algorithmic_bytes(rx, iters_per_launch, F)
        achieved = ldpc_bytes / (dec_ms * 1e-3)
        # HBM bytes of one decoder launch of this workload from the committed PMC passes (separate --pmc runs of this command,
        # tools/collect_pmc_mix.sh): (2 x FETCH_SIZE + WRITE_SIZE) KB - FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
        # gfx950's wide coalesced reads. Quoted only when the profile's stamp matches the decoder build being timed.
        mix, mix_src = profile_mix(args.decoder, headline_workload(args, F) and abs(iters_per_launch - 50.0 * F) <= 1e-6 * F)
        traffic = (2.0 * mix["FETCH_SIZE"] + mix["WRITE_SIZE"]) * 1024.0 if mix else None
        traffic_live = None
        if world == 1 and args.live_pmc:
            traffic_live = live_hbm_traffic(args)              # {kernel name: HBM bytes per launch} or None (no rocprofv3, a pass failed / timed out)
            if traffic_live is None:
                print("bench.py: --live-pmc: the rocprofv3 counter passes did not complete; roofline.traffic_live is null", file=sys.stderr, flush=True)
        issue = issue_view(mix, mix_src, dec_ms, machine, sclk)
        line = {
            "metric": ("LDPC codewords/s (rate %d/1600, max %d iters)" % (rx.K, args.iters)) if args.ldpc_only else
                      ("RX frames/s (mode %d, max %d LDPC iters)" % (args.cfg, args.iters)),
            "value": frames_total / dt, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.decoder == "spa" else "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[4]-style decoder-only soak: %d rate-%d/1600 codewords per GPU per step, noise-only LLRs, "
                                    "decoder=%s, max %d iterations" % (F, rx.K, args.decoder, args.iters)) if args.ldpc_only else
                                   ("BASELINE.json configs[1]: %d mode-%d frames per GPU per step through %s at Es/N0 %+.1f dB, "
                                    "%s variant, decoder=%s, max %d iterations" % (F, args.cfg, "2-path+AWGN" if args.channel else "AWGN",
                                                                                   args.esn0, args.variant, args.decoder, args.iters)),
                       "frames_per_step_per_gpu": F, "cfg": args.cfg, "esn0_db": args.esn0, "decoder": args.decoder,
                       "parallelism": "frame-sharded x%d, no collectives" % world},
            "ldpc_iters_per_s": iters_total / dt,
            "avg_iters_per_frame": iters_total / frames_total,
            "decoded_fraction": decoded_total / frames_total,
            "hard_frames_per_step": hard_frames / args.steps,
            "kernel_ms": {"frontend": fe_ms, "ldpc": dec_ms, "launches_averaged": nl},
            "sclk_mhz_during_run": sclk,
            "per_rank": [{"rank": r, "sclk_mhz_median": float(v[0]), "sclk_mhz_min": float(v[1]), "power_w_median": float(v[2]),
                          "ldpc_kernel_ms": float(v[3]), "frontend_kernel_ms": float(v[4]), "wall_ms": float(v[5])} for r, v in enumerate(per_rank)],
            "machine": machine,
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": mix_src, "traffic_live": traffic_live,
                         "kernel": "mgpu_ldpc_%s_kernel" % args.decoder,
                         "secondary": issue,
                         "bytes_per_codeword_iteration": b_iter,
                         "note": "algorithmic bytes (SURVEY.md 8d: 16E+4N per codeword-iteration); messages are LDS-resident so real HBM "
                                 "traffic is far lower; the decoders are bound by vector-instruction issue (see secondary), not by HBM"},
        }
        # the outputs of the last timed step, kept aside before the extras reuse the buffers: the cpu_baseline leg checks them
        S_chk = min(F, args.cpu_sample_per_core * usable_cores())
        payload_chk, stats_chk = payload[:S_chk].cpu().numpy().copy(), stats[:S_chk].cpu().numpy().copy()
        if not args.no_extras and not args.ldpc_only:
            line["extras_per_gpu"] = extras(args, rx, bufs, payload, stats, stream, dev, F, noise_amp, machine)
            # SURVEY.md §8d C2 names two points for this metric: the all-iterations one above is `value`; the realistic one (Es/N0 = threshold
            # + 3 dB, early termination at work) is this second record, same decoder, same frames per step, with its own rooflines
            opr = line["extras_per_gpu"].get("operating_point_decoder_" + args.decoder)
            if opr:
                line["operating_point"] = {"metric": "RX frames/s (mode %d at Es/N0 %+.1f dB, max %d LDPC iters, early termination)" % (args.cfg, opr["esn0_db"], args.iters),
                                           "value": opr["frames_per_s"], "unit": "frames/s", "ms_per_step": opr["ms_per_step"], "esn0_db": opr["esn0_db"],
                                           "avg_iters_per_frame": opr["avg_iters"], "ldpc_iters_per_s": opr["avg_iters"] * opr["frames_per_s"],
                                           "decoded_fraction": opr["decoded_fraction"], "kernel_ms": {"frontend": opr["frontend_ms"], "ldpc": opr["ldpc_ms"]},
                                           "decoder": args.decoder, "roofline": opr["roofline"]}
            wfr = line["extras_per_gpu"].get("waterfall_point_decoder_" + args.decoder)
            if wfr:
                line["waterfall_point"] = {"metric": "RX frames/s (mode %d at Es/N0 %+.1f dB: just below the threshold, %.0f %% of the frames run all %d iterations)"
                                                     % (args.cfg, wfr["esn0_db"], 100.0 * wfr["frames_running_all_iters"], args.iters),
                                           "value": wfr["frames_per_s"], "unit": "frames/s", "ms_per_step": wfr["ms_per_step"], "esn0_db": wfr["esn0_db"],
                                           "avg_iters_per_frame": wfr["avg_iters"], "frames_running_all_iters": wfr["frames_running_all_iters"],
                                           "ldpc_iters_per_s": wfr["avg_iters"] * wfr["frames_per_s"], "decoded_fraction": wfr["decoded_fraction"],
                                           "kernel_ms": {"frontend": wfr["frontend_ms"], "ldpc": wfr["ldpc_ms"]}, "decoder": args.decoder, "roofline": wfr["roofline"]}
        if world == 1 and not args.no_cpu_baseline and not args.ldpc_only:
            cores = usable_cores()
            S = min(F, args.cpu_sample_per_core * cores)
            last = (args.steps - 1) % nbuf
            bb_h = bufs[last][:S].cpu().numpy().view(np.complex128).reshape(S, -1)
            import oraclelib
            flags = oraclelib.FLAGS_RECEIVE_BYTE if args.variant == "receive_byte" else oraclelib.FLAGS_BASEBAND_TEST
            line["cpu_baseline"] = cpu_baseline(args.cfg, args.iters, bb_h, flags, payload_chk, stats_chk)
            if args.decoder != "spa":
                line["cpu_baseline"]["note"] = "CPU runs the reference's sum-product decoder; mismatches vs %s are expected on non-converged frames" % args.decoder
            line["cpu_baseline"]["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            # the same call through the host-buffer entry point (pageable host memory -> H2D, kernels, D2H): never `value`
            # (the whole step's batch from host memory; mgpu_rx_batch pipelines it in chunks on two streams)
            bb_all = bufs[last].cpu().numpy().view(np.complex128).reshape(F, -1)
            def median_rate(x, reps=5):                          # one warm-up call, then the median of `reps` timed calls
                out = rx.receive(x)
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    out = rx.receive(x)
                    ts.append(time.perf_counter() - t0)
                return F / sorted(ts)[reps // 2], out
            line["pcie_inclusive_frames_per_s"], out_h = median_rate(bb_all)
            hp = rx.host_path_last()       # device events of the last call: the pipeline's fill (first chunk's copy) and drain (after the last input byte)
            line["pcie_inclusive_pipeline"] = dict(hp, note="frames/s = F / (fill + steady chunks + drain); at F = %d in chunks of %d the "
                                                   "fill and drain are a fixed %.1f of the call's %.1f ms" % (F, hp["chunk_frames"], hp["fill_ms"] + hp["drain_ms"], hp["total_ms"]))
            line["pcie_inclusive_equals_device_path"] = bool(np.array_equal(out_h["payload"][:S_chk], payload_chk) and
                                                             out_h["stats"][:S_chk].tobytes() == stats_chk.tobytes())
            from mercury_amd.physical_layer import pinned_empty
            pin = pinned_empty(bb_all.shape, np.complex128)        # page-locked input (mgpu_alloc_host)
            pin[...] = bb_all
            line["pcie_inclusive_pinned_frames_per_s"], _ = median_rate(pin)
            line["pcie_bound_frames_per_s_at_55GBps"] = 55e9 / (rx.frame_samples * 16)
        emit(line, args)
    if collective:
        dist.destroy_process_group()


def dry_run(args):
    """The shard map of `bench.py --gpus N` without touching a GPU: which device every rank takes (the LOCAL_RANK rule of main(), or the pool's
    device list), that device's NUMA node when sysfs knows the amdgpu devices, and the global frame indices each rank generates and decodes
    per input batch (frame_range: SURVEY.md §8e, frame f -> rank floor(f * N / F_total))."""
    import glob
    N, F = args.gpus, args.frames
    nodes = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if os.path.basename(os.path.realpath(os.path.join(d, "driver"))) == "amdgpu":
                nodes.append({"pci_bus_id": os.path.basename(os.path.realpath(d)), "numa_node": int(open(os.path.join(d, "numa_node")).read())})
        except Exception:
            pass
    ranks = []
    for r in range(N):
        lo, hi = frame_range(r, N, F * N)
        dev = 0 if args.share_device else (r % len(nodes) if nodes else r)
        ranks.append({"rank": r, "device": dev, "placement": nodes[dev] if dev < len(nodes) else None,
                      "frames_per_step": hi - lo, "global_frames_by_input_batch": [[b * F * N + lo, b * F * N + hi] for b in range(max(1, args.nbuf))]})
    print(json.dumps({"dry_run": True, "n_gpus": N, "mode": "pool (one process, one worker thread per device)" if args.pool or N > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
                      else "ranks (torch.distributed.run, one process per GPU)", "collectives_on_the_data_path": 0,
                      "frames_per_step_total": F * N, "scaling": "weak", "amdgpu_devices_visible_in_sysfs": len(nodes), "ranks": ranks}), flush=True)


if __name__ == "__main__":
    main()
