#!/usr/bin/env python3
"""bench.py - GB/s of raw FASTQ compressed on MI355X (BASELINE.json metric), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W          (no launcher needed, any N)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the whole hot path (record index, sub-block cuts, statistics, Huffman build, packed
emit) over one GPU's share of the workload, with the FASTQ bytes already in HBM and the .ngsc
sub-block bytes left in HBM.  N=1: BASELINE configs[1], 1 GB synthetic 36 bp ERR005195-style FASTQ,
8 virtual ranks.  N>1 (weak scaling): a logical file of N x 1 GB (fixed-width variant of the same
generator so that every GPU's share can be produced on its own), 8 virtual ranks per GPU.

Two ways to run N > 1, same work per GPU, same JSON line:
  * under torch.distributed.run (WORLD_SIZE set): one process per GPU; the only collective is the all-gather of
    block sizes that assigns file offsets (RCCL over xGMI; replaces MPI_Gather/Gatherv, phyNGSC.cpp:964, 994);
  * plain `python bench.py --gpus N`: ONE process, a team (ngsc.h: ngsc_team_*): one C++ host thread per GPU calling
    ngsc_compress_device on its own device, then an RCCL all-gather of the per-rank block rows through a communicator
    created inside this process (ncclCommInitAll) - the north-star's "single-process host that partitions sub-blocks
    across the GPUs of one node ... only a tiny RCCL prefix-sum over xGMI"; config.rccl_ranks / config.collective say
    what the communicator was.
Prints ONE JSON line (rank 0 / the main thread).
"""
from __future__ import annotations

import argparse
import datetime
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from phyngsc_amd import api, fastq_gen, ngsc_format as nf  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
RANKS_PER_GPU = 8
METRIC = "GB/s raw FASTQ compressed at 1/2/4/8 MI355X; bit-exact .ngsc vs ref"


def kernel_source_hash() -> str:
    """Identifies the kernel sources a profile was taken with (profiles/traffic_latest.json carries the same hash)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "phyngsc_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(res: api.DeviceResult, raw: int) -> dict:
    """Algorithmic HBM bytes of one launch of each kernel on this workload: what the kernel must
    read of the FASTQ (or of the dense streams made from it) once, plus the stream bytes it writes.  Record tables,
    scan temporaries and code tables are NOT credited."""
    n_rec = sum(sb.n_records for sb in res.subblocks)
    bases = sum(sb.n_bases for sb in res.subblocks)
    title_in = raw - 2 * bases - 4 * n_rec   # title lines incl. their newline
    title_out = sum(sb.title_len for sb in res.subblocks)
    q_out = sum(sb.qual_len for sb in res.subblocks)
    d_out = sum(sb.dna_len for sb in res.subblocks)
    return {
        "k_nl_count": raw,
        "k_nl_write": raw // 8,
        "k_title_stats_swar": title_in,
        "k_title_stats_reg": title_in,
        "k_title_measure_swar": 0,
        "k_title_measure_reg": title_in,
        "k_title_pack_reg": title_out,
        "k_title_hist": title_in,
        "k_sq_stats_reg": 2 * bases,
        "k_sq_measure_reg": bases,
        "k_sq_pack_reg": 2 * bases + q_out + d_out,
        "k_seq_qual_stats": 2 * bases,
        "k_measure_sq": 2 * bases,
        "k_pack_sq": 2 * bases + q_out + d_out,
        "k_zero_output": sum(sb.size for sb in res.subblocks),
    }


def cpu_baseline(data: np.ndarray, P: int, allow_port: bool = True) -> dict:
    """Reported-only CPU baseline on the host cores of the GPU box: the unmodified reference binary
    (oracle/_ref, built in the build container) if it runs here, else the oracle port (with the reason).
    P = MPI ranks (one thread each, as the goldens are made); allow_port False: the reference or nothing."""
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "phyNGSC")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    sample = f"whole workload ({data.size} B), page-cached file"
    why = None
    if not os.path.exists(ref):
        why = "oracle/_ref/phyNGSC not present"
    elif not os.path.exists(mpiexec):
        why = "no mpiexec on this box"
    else:
        try:
            with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
                fq, out = os.path.join(tmp, "in.fastq"), os.path.join(tmp, "out.ngsc")
                data.tofile(fq)
                npn = min(P, cores)
                t0 = time.time()
                r = subprocess.run([mpiexec, "-np", str(npn), ref, fq, out, "1"], capture_output=True, text=True, timeout=600)
                dt = time.time() - t0
                if r.returncode == 0 and os.path.getsize(out) > 0:
                    times = [float(l.split()[1]) for l in r.stdout.splitlines() if l[:1].isdigit() and len(l.split()) >= 4]
                    comp = max(times) if times else dt
                    return {"value": data.size / comp / 1e9, "unit": "GB/s", "cores": npn, "host_cores": cores,
                            "kind": "reference",
                            "sample": sample + f"; mpiexec -np {npn} x 1 thread, slowest-rank COMP_TIME {comp:.2f} s "
                                               f"(wall {dt:.2f} s)"}
                why = f"reference run failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]!r}"
        except Exception as e:  # noqa: BLE001
            why = f"reference run raised {type(e).__name__}: {e}"
    if not allow_port:
        return {"value": None, "unit": "GB/s", "cores": 0, "host_cores": cores, "kind": "reference", "fallback_reason": why}
    from oracle import binding as orc
    thr = min(P, cores)
    _, st = orc.compress_file(data, P, threads=thr)
    return {"value": data.size / st.seconds / 1e9, "unit": "GB/s", "cores": thr, "host_cores": cores, "kind": "port",
            "fallback_reason": why,
            "sample": sample + f"; oracle/ngsc_oracle.c, {thr} OpenMP threads over {P} virtual ranks"}


class Share:
    """One GPU's share of the workload: its byte range of the (logical) file in HBM, its virtual ranks, its context."""

    def __init__(self, world: int, rank: int, device: int, per_gpu: int):
        self.world, self.rank, self.device = world, rank, device
        self._stream = None
        self.P = RANKS_PER_GPU * world
        P = self.P
        if world == 1:
            cfg = fastq_gen.CFG2
            host = fastq_gen.generate(cfg, target_bytes=per_gpu)
            self.file_size, self.buf_off = int(host.size), 0
            self.workload = f"cfg2: {host.size} B synthetic FASTQ, 36 bp ERR005195-style (seed 2), {P} virtual ranks, 1 GPU"
        else:
            cfg = fastq_gen.CFG2_FIXED
            rs = fastq_gen.record_size(cfg)
            n_per = per_gpu // rs
            self.file_size = world * n_per * rs
            lo, hi = api.job_byte_range(self.file_size, P, rank * RANKS_PER_GPU, (rank + 1) * RANKS_PER_GPU)
            self.buf_off = lo & ~15
            host = fastq_gen.generate_range(cfg, self.buf_off, hi - self.buf_off)
            self.workload = (f"cfg2-fixed: {self.file_size} B logical FASTQ ({world} x {n_per * rs} B), 36 bp, {P} virtual "
                             f"ranks, {RANKS_PER_GPU} per GPU")
        self.host = host
        self.lo_rank, self.hi_rank = rank * RANKS_PER_GPU, (rank + 1) * RANKS_PER_GPU
        fs = self.file_size
        self.my_raw = (api.region_bounds(fs, P, self.hi_rank - 1)[1] + 1 if self.hi_rank < P else fs) - \
            api.region_bounds(fs, P, self.lo_rank)[0]
        if self.hi_rank < P:
            self.my_raw -= api.OVERLAP  # count each byte once: regions, not regions + overlap
        torch.cuda.set_device(device)
        dev = torch.device("cuda", device)
        self.d_in = torch.from_numpy(host).to(dev)
        self.d_out = torch.empty(int(host.size * 0.45) + (32 << 20), dtype=torch.uint8, device=dev)
        self.eng = api.Engine(device)
        self.res = None

    def step(self):
        if self._stream is None:   # (looked up once: a Stream object per step is 5-10 us of Python on a 2 ms step)
            self._stream = torch.cuda.current_stream(self.d_in.device).cuda_stream
        self.res = self.eng.compress_device(self.d_in, self.file_size, self.P, rank_lo=self.lo_rank, rank_hi=self.hi_rank,
                                            buf_off=self.buf_off, out=self.d_out, stream=self._stream)
        return self.res

    def block_sizes(self):
        res = self.res
        sizes = [[b.size for b in api.plan_blocks(self.P, rk.rank, [res.subblocks[rk.first_subblock + k].size
                                                                    for k in range(rk.n_subblocks)])] for rk in res.ranks]
        meta = [(rk.n_subblocks, rk.overlap_used) for rk in res.ranks]
        return sizes, meta


def other_configs(eng, size: int = 400_000_000, steps: int = 5) -> dict:
    """The BASELINE configurations the headline is NOT quoted on, measured the same way on 400 MB each, outside the headline's
    timed region: configs[2]'s shape (150-base reads, 0.2 % N bases) and configs[4]'s (36-300 bases, skewed qualities, long
    titles).  Wall time of `steps` calls after two warm-up calls; dominant kernel from an untimed pass with every kernel
    timed alone; frac = that kernel's algorithmic bytes / its launch time / 8 TB/s."""
    out = {}
    for name, cfg in (("cfg3_150bp", fastq_gen.CFG3), ("cfg5_36_300", fastq_gen.CFG5_300)):
        data = fastq_gen.generate(cfg, target_bytes=size)
        d_in = torch.from_numpy(data).cuda()
        d_out = torch.empty(int(data.size * 0.6) + (32 << 20), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            res = eng.compress_device(d_in, data.size, RANKS_PER_GPU, out=d_out)
        eng.profile_only(None)
        eng.set_profiling(True)
        eng.reset_kernel_times()
        res = eng.compress_device(d_in, data.size, RANKS_PER_GPU, out=d_out)
        torch.cuda.synchronize()
        kt = eng.kernel_times()
        eng.set_profiling(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = eng.compress_device(d_in, data.size, RANKS_PER_GPU, out=d_out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        per = {k: ms / max(n, 1) for k, (ms, n) in kt.items() if n}
        dom = max(per, key=per.get)
        alg = algorithmic_bytes(res, data.size)
        out[name] = {"GBps": round(data.size / dt / 1e9, 1), "ms": round(dt * 1e3, 3), "raw_bytes": int(data.size),
                     "ratio": round(res.total_bytes / data.size, 4), "dominant_kernel": dom, "dominant_ms": round(per[dom], 4),
                     "frac": round(alg.get(dom, 0) / (per[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "kernels_ms": {k: round(v, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]}}
        del d_in, d_out
    return out


def profile_pass(sh: Share):
    """Untimed: an event pair around EVERY kernel -> the per-kernel table and which kernel dominates; then only that
    kernel keeps its events for the timed region (an event pair around each launch costs launch gaps)."""
    eng = sh.eng
    eng.set_profiling(True)
    eng.reset_kernel_times()
    for _ in range(2):
        sh.step()
    torch.cuda.synchronize(sh.device)
    ktimes_all = eng.kernel_times()
    dom = max(ktimes_all, key=lambda k: ktimes_all[k][0] / max(ktimes_all[k][1], 1))
    eng.profile_only(dom)
    eng.reset_kernel_times()
    return ktimes_all, dom


def report(sh: Share, args, world, dt, total_raw, total_out, ktimes_all, dom, launch_mode, extra_config=None):
    res, eng = sh.res, sh.eng
    ktimes = eng.kernel_times()
    alg = algorithmic_bytes(res, sh.my_raw)
    per_kernel = {k: {"ms_avg": ms / max(n, 1), "launches": n} for k, (ms, n) in ktimes_all.items() if n}
    dom_ms = ktimes[dom][0] / max(ktimes[dom][1], 1)   # measured inside the timed region
    per_kernel[dom]["ms_avg"] = dom_ms
    achieved = alg.get(dom, 0) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM traffic of the dominant kernel: NOT measured in this run (PMC counters need their own rocprofv3 passes):
    # taken from the committed profile, with its provenance, and marked stale when the kernel sources changed since
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if world == 1 and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        traffic = tj.get(dom, {}).get("hbm_bytes_per_launch")
        prof_hash = tj.get("kernel_source_hash")
        traffic_src = {"file": "profiles/traffic_latest.json", "tool": "tools/pmc_traffic.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                       "separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md)", "build": tj.get("build"),
                       "measured_in_this_run": False, "stale": prof_hash != kernel_source_hash()}
    kernel_sum_ms = sum(v["ms_avg"] for v in per_kernel.values())
    step_ms = dt / args.steps * 1e3
    line = {
        "metric": METRIC,
        "value": total_raw * args.steps / dt / 1e9,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": sh.workload, "virtual_ranks": sh.P, "raw_bytes": total_raw, "ngsc_bytes": total_out,
                   "ratio": total_out / total_raw, "subblocks_rank0": len(res.subblocks), "launch": launch_mode},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg.get(dom, 0), "avg_launch_ms": dom_ms},
        # kernel_ms_sum: every kernel timed ALONE (the untimed profile pass runs the serial schedule); the timed steps run
        # the default two-stream schedule, in which the latency-bound kernels hide beside the throughput kernels, so a step
        # can be shorter than the sum (serial_minus_step_ms > 0); NGSC_OVERLAP=0 times the serial schedule
        "pipeline": {"kernel_ms_sum": kernel_sum_ms, "serial_minus_step_ms": kernel_sum_ms - step_ms,
                     "schedule": "serial" if os.environ.get("NGSC_OVERLAP") == "0" else
                                 ("two chains" if os.environ.get("NGSC_OVERLAP") == "1" else "staggered two-stream"),
                     "raw_GBps_over_kernel_time": sh.my_raw / (kernel_sum_ms * 1e-3) / 1e9 if kernel_sum_ms else None,
                     "frac_of_hbm_read_roofline": total_raw / world * args.steps / dt / 1e9 / HBM_PEAK_GBS},
        # the same roofline figure for the other throughput kernels, each timed ALONE in the untimed profile pass (the
        # dominant one: inside the timed region): algorithmic bytes / launch duration / 8 TB/s
        "roofline_by_kernel": {k: {"ms": round(v["ms_avg"], 4), "algorithmic_bytes": alg[k],
                                   "frac": round(alg[k] / (v["ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                               for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms_avg"])[:8]
                               if alg.get(k) and v["ms_avg"] > 0},
        "kernels_ms": {k: round(v["ms_avg"], 4) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["ms_avg"])},
    }
    if extra_config:
        line["config"].update(extra_config)
    if world > 1:
        # the per-kernel tables of an N > 1 line are rank 0's GPU, profiled by itself in an untimed pass: they say nothing about
        # contention on the other GPUs or on the host; the line's value and ms_per_step are the max-over-ranks wall time
        line["config"]["kernel_tables_scope"] = "GPU of rank 0 alone, untimed pass (not the timed multi-GPU steps)"
    if world == 1 and not args.no_verify:
        out = api.assemble_container(res, sh.P, sh.file_size)
        man_path = os.path.join(ROOT, "tests", "golden", "manifest_cfg2_1GB.json")
        verdict = None
        if os.path.exists(man_path):
            with open(man_path) as f:
                man = json.load(f)
            if man["fastq_bytes"] == sh.file_size and str(sh.P) in man["ranks"]:
                verdict = nf.manifest(out) == man["ranks"][str(sh.P)]
        line["bit_exact_vs_reference_manifest"] = verdict
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sh.host, sh.P)
        # a second, wider point (reported only): the same reference binary with one MPI rank per core up to 64.  A different
        # rank count gives a different - equally valid - container, so this is a timing, not a parity run.
        wide = min(64, os.cpu_count() or 1)
        line["cpu_baseline_wide"] = cpu_baseline(sh.host, wide, allow_port=False) if wide > sh.P else None
    else:
        line["cpu_baseline"] = None
    if world == 1 and not args.no_other_configs:
        line["other_configs"] = other_configs(eng)
    print(json.dumps(line), flush=True)


# ---- one process per GPU (torch.distributed.run) ----------------------------------------------------------------
def run_distributed(args, world: int):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NGSC_BENCH_REHEARSE=1: every rank on cuda:0 with gloo collectives - a dry run of the N > 1 path on a one-GPU box
    # (RCCL refuses two ranks on one device); never used by the driver
    rehearse = os.environ.get("NGSC_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev   # where the (tiny) collective tensors live
    pg = None                                            # the RCCL group of the exchange (None: the default group)
    coll = "gloo rehearsal on one device" if rehearse else "RCCL all-gather of block sizes"
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            # gloo is the agreement channel (barriers, and the decision below); the block-size exchange itself goes over an
            # RCCL group when EVERY rank got one.  The decision is collective: a rank that fell back on its own would
            # leave its peers waiting inside an RCCL call (a rank stuck there comes back with the group's timeout).
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
            ok, reason = 1, ""
            if os.environ.get("NGSC_BENCH_COLL") == "gloo":
                ok, reason = 0, "NGSC_BENCH_COLL=gloo"
            else:
                try:
                    pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                    t = torch.ones(1, device=dev)
                    dist.all_reduce(t, group=pg)
                    torch.cuda.synchronize()
                    if int(t.item()) != world:
                        raise RuntimeError(f"all_reduce of ones gave {t.item()}")
                except Exception as e:   # noqa: BLE001
                    ok, reason = 0, f"{type(e).__name__}: {e}".splitlines()[0][:160]
            flag = torch.tensor([ok], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                pg, cdev = None, torch.device("cpu")
                coll = f"gloo all-gather of block sizes (RCCL unusable on some rank{': ' + reason if reason else ''})"
    sh = Share(world, rank, local_rank, args.size_mb * 1_000_000)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sh.step()
    ktimes_all, dom = profile_pass(sh)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sh.step()
    barrier()
    dt = time.perf_counter() - t0
    sh.eng.set_profiling(False)
    sh.eng.profile_only(None)
    if world > 1:
        # the one (data-path-free) collective: block sizes -> file offsets (RCCL all-gather over xGMI)
        sizes_local, meta_local = sh.block_sizes()
        sizes_all, _ = api.exchange_block_sizes(sizes_local, meta_local, dist, device=cdev, group=pg)
        api.file_offsets(sizes_all, api.ranks_per_group(sh.file_size, sh.P))
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
        dt = float(t.item())
        tot = torch.tensor([sh.my_raw, sh.res.total_bytes], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=pg)
        total_raw, total_out = int(tot[0].item()), int(tot[1].item())
    else:
        total_raw, total_out = sh.file_size, sh.res.total_bytes
    if rank == 0:
        mode = ("one process per GPU (torch.distributed.run), " + coll) if world > 1 else "one process, one context"
        extra = None
        if world > 1:
            extra = {"rccl_ranks": dist.get_world_size(pg) if pg is not None else 0, "collective": coll, "collective_per_step": False}
        report(sh, args, world, dt, total_raw, total_out, ktimes_all, dom, mode, extra)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---- one process, N GPUs: a team (one host thread per GPU inside the library + an in-process RCCL communicator) ----------
def run_team(args):
    """`python bench.py --gpus N` without a launcher: the north star's single-process host.  Every step is ONE call of
    ngsc_team_compress: each GPU compresses its 8 virtual ranks on its own host thread (C++, no GIL), then the per-rank block
    rows are all-gathered with RCCL over xGMI (ncclCommInitAll inside this process) and every GPU holds what it needs for
    the file offsets.  config.rccl_ranks is what the communicator reports; 0 means the rows met in host memory and
    config.collective says why."""
    n = args.gpus
    ndev = torch.cuda.device_count()
    rehearse = os.environ.get("NGSC_BENCH_REHEARSE") == "1"   # all contexts on device 0 (one-GPU box): no RCCL there
    if ndev < n and not rehearse:
        raise SystemExit(f"--gpus {n} but this node shows {ndev} GPU(s)")
    shares = [Share(n, i, 0 if rehearse else i, args.size_mb * 1_000_000) for i in range(n)]
    team = api.Team([sh.eng for sh in shares])
    sh0 = shares[0]
    P, fs = sh0.P, sh0.file_size
    row_len = 3 + int(fs // P * 0.6 // (8 << 20)) + 4
    shards = [(sh.d_in, sh.buf_off, sh.lo_rank, sh.hi_rank, sh.d_out) for sh in shares]

    def sync_all():
        for sh in shares:
            torch.cuda.synchronize(sh.device)

    rows, ms = None, None
    for _ in range(args.warmup):
        rows, ms = team.compress(shards, fs, P, row_len)
    ktimes_all, dom = profile_pass(sh0)         # per-kernel table from GPU 0 alone (untimed)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows, ms = team.compress(shards, fs, P, row_len)
    sync_all()
    dt = time.perf_counter() - t0
    # the offsets every GPU derives from the gathered rows, against the Python statement of the same order
    sizes_all, _ = api.sizes_from_rows(rows)
    assert api.rows_to_offsets(rows, fs) == api.file_offsets(sizes_all, api.ranks_per_group(fs, P))
    sh0.res = sh0.step()                        # (report() reads GPU 0's result tables)
    total_raw = sum(sh.my_raw for sh in shares)
    total_out = sum(int(sh.eng._L.ngsc_result_bytes(sh.eng._h)) for sh in shares)
    sh0.eng.set_profiling(False)
    mode = f"one process, team of {n} host threads / contexts" + (" (rehearsal: all on device 0)" if rehearse else "")
    # the exchange by itself (every step ends in it): the same rows through the bare collective, timed apart from the compression
    t1 = time.perf_counter()
    for _ in range(20):
        team.allgather_u64(rows.reshape(n, -1))
    coll_ms = (time.perf_counter() - t1) / 20 * 1e3
    extra = {"rccl_ranks": team.rccl_ranks, "collective": team.describe, "collective_per_step": True,
             "collective_ms_per_step": round(coll_ms, 4), "member_ms_last_step": [round(x, 3) for x in ms],
             "member_exchange_ms_last_step": [round(x, 3) for x in team.exchange_ms()]}
    report(sh0, args, n, dt, total_raw, total_out, ktimes_all, dom, mode, extra)
    team.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size-mb", type=int, default=1000, help="raw FASTQ MB per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 400 MB runs of the 150-base and 36-300-base configurations")
    args = ap.parse_args()
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        run_distributed(args, world)
    elif args.gpus == 1:
        run_distributed(args, 1)
    else:
        run_team(args)


if __name__ == "__main__":
    main()
