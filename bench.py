#!/usr/bin/env python3
"""bench.py -- throughput of the quantized hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...      (starts N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): the synthetic INT8-weight x FP16-activation matmul sweep,
K, N in {1024, 2048, 4096, 8192}.  One *step* is one pass of the decode-shaped (M = 1) sweep:
for each of the 16 (K, N) shapes the FP16-activation matmul of compute_matmul_node
(graph_ops_nn.cpp:166-177: row quantization + grouped INT8 matmul, kernel_gemm.cpp:128-327),
FP16 in / FP16 out, on weights already resident in HBM.  The 16 matmuls of a step are
independent, and the library takes them as one batch (cactus_hip_matmul_int8_f16act_batch_dev):
one launch whose workgroups each quantize their activation row and stream a 16-row weight tile
of one of the shapes.  Four complete weight sets (--weight-sets; 229 MiB each) take turns
between steps so that no weight is served from the 256 MiB Infinity Cache.  `value` is
algorithmic bytes of a step (weights + scales + activations + outputs) times ranks, divided by
the slowest rank's time per step.

Ranks are independent replicas (independent sequences): no collective on the data path, weak
scaling.  The same JSON line carries
  roofline      -- the dominant kernel (8192 x 8192 INT8 GEMV) timed alone with HIP events
  cpu_baseline  -- the CPU restatement of the reference kernel on the same sweep (rank 0, N=1)
  extra         -- per-shape GEMV rates (INT8 and packed INT4), M = 16, M = 512 MFMA rate,
                   decode attention over the INT8 KV cache.
"""
import argparse
import threading
import subprocess
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIZES = [1024, 2048, 4096, 8192]
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E, spec
INT8_MFMA_PEAK_TOPS = 5000.0  # dense INT8 = 2x the ~2.5 PF dense BF16 rate


def gemv_bytes(M, K, N, int4=False, fused_quant=True):
    """Algorithmic bytes of one matmul call (weights + scales + activations + outputs)."""
    w = N * K // 2 if int4 else N * K
    act = 2 * M * K if fused_quant else (M * K + 4 * M)
    return w + 2 * N * (K // 128) + act + 2 * M * N


def make_weights(K, N, int4, gen, pack=False):
    """Row-major weights as the reference's files hold them; pack: also the tile-major twin the library keeps for
    the decode-shaped launches (cactus_hip_weights_pack_dev, made once when a weight becomes resident)."""
    if int4:
        B = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device="cuda", generator=gen)
    else:
        B = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=gen)
    Bs = (0.01 + torch.rand((N, K // 128), device="cuda", generator=gen) * 0.05).half()
    if pack:
        import cactus_amd.hip as H
        H.pack_weights(B)
    return B, Bs


def time_events(fn, iters, warm=3):
    """Average milliseconds per call of fn(i), HIP events on torch's current stream (the stream
    the kernels are launched on)."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_graph(fn, n_calls, replays=6):
    """Average milliseconds per call of fn(i), i in range(n_calls), with the n_calls launches
    captured into one HIP graph (no host launch cost between them) and replayed."""
    for i in range(min(n_calls, 3)):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n_calls):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (replays * n_calls)


def cpu_baseline(work, steps=5):
    """The oracle (CPU restatement of the reference kernel -- its NEON sources do not build on
    x86) on the same M=1 sweep and the SAME buffers the GPU step reads (weight set 0 and the
    activation rows, copied back to the host), with the reference's thread policy for M <= 1: 4 GEMM
    threads (kernel_utils.h:450-455); 1 warm-up + `steps` timed passes (tests/test_threading.cpp:111-120).
    work: [(x fp16 [1, K], B int8 [N, K], B_scales fp16 [N, K/128])] as numpy arrays."""
    from oracle import binding as O
    threads = min(4, os.cpu_count() or 1)
    O.set_threads(threads)
    total = sum(gemv_bytes(1, B.shape[1], B.shape[0]) for _, B, _ in work)

    def one_pass():
        for x, B, Bs in work:
            A, As = O.quantize_activations(x)
            O.matmul_int8(A, As, B, Bs, 128)

    one_pass()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    dt = (time.perf_counter() - t0) / steps
    O.set_threads(1)
    return {"value": round(total / dt / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"full M=1 INT8 sweep on the GPU step's own buffers (16 shapes, {total / 1e6:.0f} MB/pass), "
                      f"1 warm-up + {steps} timed passes, {threads} threads = the reference's GEMM policy for M<=1; "
                      "scalar C restatement compiled with gcc -O3 -march=x86-64-v3 (NEON reference is unbuildable on x86)",
            "ms_per_pass": round(dt * 1e3, 2)}


def live_traffic(kernel="skinny_batch"):
    """HBM bytes per launch of the timed kernel, measured now: this same command (no extras, no CPU leg, 3 steps)
    run as a child under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (one counter per pass,
    nothing else traced).  FETCH_SIZE is in KiB and, on gfx950, tallies the 128-byte requests of wide coalesced
    reads at 64 bytes each: x 1024 x 2; WRITE_SIZE x 1024.  None when rocprofv3 is absent or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    means = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["timeout", "-s", "KILL", "75", exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--traffic-child", "--no-extra", "--no-cpu",
                   "--steps", "3", "--warmup", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=100)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter]
            if not vals:
                return None
            means[counter] = (sum(vals) / len(vals), len(vals))
    except (OSError, subprocess.SubprocessError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = int(means["FETCH_SIZE"][0] * 1024 * 2), int(means["WRITE_SIZE"][0] * 1024)
    return {"hbm_bytes_per_launch": rd + wr,
            "source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (%d launches; KiB x 1024, "
                      "reads x 2 for gfx950)" % means["FETCH_SIZE"][1]}


def greedy_identity_summary():
    """Free-running greedy identity, as measured by scripts/greedy_identity.py on the GPU box (64 device-only steps per
    run, the CPU restatement shadowing the device's token stream; tests/test_gpu_free_running.py asserts the same
    bars): the committed table profiles/r03_greedy_identity.json condensed per case.  The bench itself runs no CPU
    decode (only the cpu_baseline leg may call the restatement)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_greedy_identity.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    out = {"source": "profiles/r03_greedy_identity.json (scripts/greedy_identity.py --full; 64 free-running steps per run)"}
    for case, rows in table.items():
        runs = list(rows.values())
        diffs = [d for r in runs for d in r["different_choices"]]
        out[case] = {"runs": len(runs), "runs_with_all_64_identical": sum(r["identity_length"] == r["steps"] for r in runs),
                     "shortest_identity_length": min(r["identity_length"] for r in runs),
                     "differing_choices": len(diffs),
                     "largest_oracle_margin_at_a_differing_choice": max((d["oracle_margin"] for d in diffs), default=0.0),
                     "largest_margin_over_logit_error_at_a_differing_choice":
                         round(max((d["oracle_margin"] / d["logit_error"] for d in diffs), default=0.0), 3),
                     "worst_logit_error_rel_to_largest_logit": max(r["worst_logit_error_rel"] for r in runs)}
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this file with torch.distributed.run as
    a child process (one rank per GPU, rendezvous on 127.0.0.1), pass its output through (rank 0 prints the one
    JSON line) and return its exit code -- non-zero if any rank failed.  The parent makes no GPU call."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_extra:
        cmd.append("--no-extra")
    if args.no_cpu:
        cmd.append("--no-cpu")
    print("[bench] launching " + " ".join(cmd[1:]), file=sys.stderr, flush=True)
    return subprocess.run(cmd).returncode


EXTRAS_LIMIT_S = 420     # several ranks: time the additions to the headline may take before the run ends itself


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--weight-sets", type=int, default=4, help="complete weight sets the timed steps use in turn (240 MB each)")
    ap.add_argument("--no-extra", action="store_true", help="skip the per-shape / M=16 / M=512 / attention table")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # the run inside a --pmc pass
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: this process becomes the launcher of N ranks and never touches a GPU
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    live = None
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or \
        any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)     # already under a profiler: no nested one
    if world == 1 and not args.no_traffic and not args.traffic_child and not profiled:
        live = live_traffic()        # child processes under rocprofv3, before this process touches the GPU
    if world > 1:
        print(f"[bench] rank {rank}/{world} (local rank {local_rank}, pid {os.getpid()}) starting", file=sys.stderr, flush=True)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    import cactus_amd.hip as H
    H.init(torch.cuda.current_device())

    gen = torch.Generator(device="cuda").manual_seed(42 + rank)

    # ---- workload: NS complete weight sets used in turn (more than the 256 MB Infinity Cache holds), FP16 activation rows per K
    sets = []
    NS = max(2, args.weight_sets)
    for _ in range(NS):
        sets.append({(K, N): make_weights(K, N, False, gen) for K in SIZES for N in SIZES})
    acts = {K: (torch.rand((1, K), device="cuda", generator=gen) * 2 - 1).half() for K in SIZES}
    outs = {(K, N): torch.empty((1, N), dtype=torch.float16, device="cuda") for K in SIZES for N in SIZES}
    step_bytes = sum(gemv_bytes(1, K, N) for K in SIZES for N in SIZES)
    # big shapes first: their workgroups start streaming while the small ones fill the gaps
    order = sorted(((K, N) for K in SIZES for N in SIZES), key=lambda kn: -kn[0] * kn[1])
    batches = [H.MatmulBatch([(acts[K], ws[(K, N)][0], ws[(K, N)][1], outs[(K, N)]) for K, N in order], 128)
               for ws in sets]

    def step(i, batched=True):
        if batched:
            batches[i % NS].run()
        else:
            ws = sets[i % NS]
            for K, N in order:
                H.matmul_f16act(acts[K], ws[(K, N)][0], ws[(K, N)][1], 128, out=outs[(K, N)])

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # Steps are captured into HIP graphs (the kernels allocate nothing and run on the capturing
    # stream): CHUNK consecutive steps per graph, as a decode loop captures a whole token's
    # launches, plus single-step graphs for the remainder.  Weight sets alternate step by step.
    CHUNK = 12 if NS in (2, 3, 4, 6) else 2 * NS      # a multiple of NS: every chunk starts on set 0
    for i in range(NS):
        step(i)
    torch.cuda.synchronize()

    def capture(batched, nsteps=1, first=0):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for j in range(nsteps):
                step(first + j, batched)
        return g

    # Every weight gets its tile-major twin, as a host does once when it loads a model; all timed steps stream the
    # twins (extra.gemv_sweep_row_major_layout times the same step on weights without twins).
    for ws in sets:
        for B, _ in ws.values():
            H.pack_weights(B)
    for i in range(NS):
        step(i)
    torch.cuda.synchronize()
    graphs = [capture(True, 1, i) for i in range(NS)]
    chunk_graph = capture(True, CHUNK, 0)

    def run_step(i):
        graphs[i % NS].replay()

    def run_steps(n):
        for _ in range(n // CHUNK):
            chunk_graph.replay()
        for i in range(n % CHUNK):
            run_step(i)

    run_steps(args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = step_bytes * world / (dt / args.steps) / 1e9

    # the same step as 16 separate launches (one fused call per shape, same stream)
    seq_graph = capture(False, 2, 0)
    seq_ms = time_events(lambda i: seq_graph.replay(), 6, warm=2) / 2

    # ---- roofline: the timed region holds one kernel, the batched GEMV launch (one per step):
    # its average duration from HIP events on the launching stream around replays of the
    # CHUNK-step graph (kernels back to back).
    ev_ms = time_events(lambda i: chunk_graph.replay(), 4, warm=1) / CHUNK
    achieved = step_bytes / (ev_ms * 1e-3) / 1e9
    # HBM bytes per launch of this kernel from the PMC pass of this same command (collected in
    # its own rocprofv3 run, summarised by scripts/pmc_traffic.py); null when no summary exists.
    traffic, traffic_source = None, None
    if live is not None:
        traffic, traffic_source = live["hbm_bytes_per_launch"], live["source"]
    else:   # no profiler here (or a pass failed): the committed summary of the same command, named as such
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f)
            if str(pm.get("kernel", "")).startswith("skinny_batch") and \
                    pm.get("algorithmic_bytes_per_launch", step_bytes) == step_bytes:
                traffic, traffic_source = pm.get("hbm_bytes_per_launch"), "profiles/pmc_traffic.json (an earlier run of this command)"
        except (OSError, ValueError):
            pass
    roofline = {"bound": "hbm", "kernel": "skinny_batch_sub_kernel<INT8> (16 fused row-quant GEMVs, M=1, one launch, tile-major weights)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": step_bytes, "us_per_launch": round(ev_ms * 1e3, 2)}
    # the largest single shape launched alone (8 rotating weight buffers, 512 MB > Infinity Cache)
    K = N = 8192
    big = [sets[0][(K, N)], sets[1][(K, N)]] + [make_weights(K, N, False, gen, pack=True) for _ in range(6)]
    o = outs[(K, N)]
    x8 = acts[K]
    ms = time_events(lambda i: H.matmul_f16act(x8, big[i % 8][0], big[i % 8][1], 128, out=o), 80, warm=8)
    kbytes = gemv_bytes(1, K, N)
    single = {"kernel": "skinny_kernel<INT8, 8 waves, fused row quant> (GEMV, M=1, K=N=8192, one launch)",
              "us_per_launch": round(ms * 1e3, 2), "GBps": round(kbytes / (ms * 1e-3) / 1e9, 1),
              "frac_hbm": round(kbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "bytes_per_launch": kbytes}

    line = None
    if rank == 0:
        line = {
            "metric": "quant_gemv_hbm_throughput", "value": round(value, 1), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": "INT8-weight x FP16-activation matmul sweep, M=1, K,N in {1024,2048,4096,8192} "
                                   "(16 shapes/step: fused row quant + grouped INT8 GEMV, group 128); "
                                   f"{NS} weight sets used in turn ({NS * 229} MiB); one step = one batched launch, "
                                   f"{CHUNK} steps per HIP graph; weights resident row-major [N,K] with the "
                                   "library's tile-major twins (cactus_hip_weights_pack_dev, made once at load)",
                       "bytes_per_step": step_bytes,
                       "separate_launches_ms_per_step": round(seq_ms, 4),
                       "parallelism": f"replicas x{world} (independent sequences)"},
            "roofline": roofline,
        }

    # The headline is measured; what follows only adds to the line.  With several ranks the additions run code whose
    # collectives involve every rank (RCCL through the library's own communicator): if a rank fails there the others
    # would wait for it for ever, so every rank arms a watchdog that ends the run cleanly -- rank 0 prints the line with
    # what it has -- once the additions have taken longer than EXTRAS_LIMIT_S.
    printed = threading.Event()
    emit_lock = threading.Lock()

    def emit(note=None):
        with emit_lock:
            if rank == 0 and not printed.is_set():
                if note:
                    line.setdefault("extra", {})["multi_gpu_extras"] = note
                print(json.dumps(line), flush=True)
                printed.set()

    watchdog = None
    if world > 1:
        def expire():
            emit(f"not finished after {EXTRAS_LIMIT_S} s: the run was ended by its watchdog")
            os._exit(0)
        watchdog = threading.Timer(EXTRAS_LIMIT_S, expire)
        watchdog.daemon = True
        watchdog.start()

    extra = {}
    if not args.no_extra and rank == 0 and world == 1:   # the secondary table is a single-GPU measurement
        extra = extra_table(H, gen, big)
    if rank == 0:
        extra = dict(extra, single_launch_8192=single)
        gi = greedy_identity_summary()
        if gi is not None:
            extra["greedy_identity"] = gi
        line["extra"] = extra
    if world > 1 and not args.no_extra:
        try:
            sd = sharded_decode(H, gen, world)
            if rank == 0:
                line["extra"]["sharded_decode"] = sd
            ns = nsplit_8b(H, world, rank)
            if rank == 0:
                line["extra"]["nsplit_8b"] = ns
        except Exception as e:     # the other ranks may now wait in a collective: the watchdogs end the run
            print(f"[bench] rank {rank}: multi-GPU extras failed: {e!r}", file=sys.stderr, flush=True)
            if rank == 0:
                line["extra"]["multi_gpu_extras"] = f"failed on rank 0: {e!r}"

    if rank == 0:
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline([(acts[K].cpu().numpy(), sets[0][(K, N)][0].cpu().numpy(),
                                                  sets[0][(K, N)][1].cpu().numpy()) for K, N in order])
            line["extra"] = line.pop("extra")      # keep the order of the keys: extra last
        emit()
    if world > 1:
        import torch.distributed as dist
        try:
            dist.destroy_process_group()
        except Exception as e:
            print(f"[bench] rank {rank}: destroy_process_group: {e!r}", file=sys.stderr, flush=True)
        watchdog.cancel()


def sharded_decode(H, gen, world):
    """Independent sequences sharded over the ranks, no collective in the data path: every rank runs
    whole decode steps on its own sequences (full weight replica, its own KV cache); the job's rate is
    the sequences of all ranks over the slowest rank's step time.  Qwen3-0.6B: 256 sequences per GPU
    (the single-GPU configuration repeated per rank); LFM2-1.2B shape: 1024 sequences divided over the
    ranks.  Every rank reaches the one all-reduce whatever happened before it."""
    import torch.distributed as dist
    ms = [1e30, 1e30]
    try:
        ms[0] = qwen3_decode_step(H, gen, 256, 512)["ms_per_step"]
        torch.cuda.empty_cache()
    except Exception as e:     # reported as a missing row, never as a hang
        print(f"sharded_decode: qwen3 step failed on a rank: {e}", file=sys.stderr)
    per_rank = max(1024 // world, 1)
    try:
        ms[1] = lfm2_decode_step(H, gen, per_rank, 512, mixed=True)["ms_per_step"]     # configs[3]: mixed precision
        torch.cuda.empty_cache()
    except Exception as e:
        print(f"sharded_decode: lfm2 step failed on a rank: {e}", file=sys.stderr)
    tt = torch.tensor(ms, dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    q_ms, l_ms = (float(v) for v in tt.tolist())
    out = {"scaling_note": "max step time over ranks; qwen3 weak (256 sequences per GPU), lfm2 strong (1024 sequences in all)"}
    if q_ms < 1e29:
        out["qwen3_0p6b_b256_per_gpu_c512"] = {"ms_per_step": round(q_ms, 4), "sequences": 256 * world,
                                                "tokens_per_s": round(256 * world / (q_ms * 1e-3), 1)}
    if l_ms < 1e29:
        out["lfm2_1p2b_mixed_1024_sequences_c512"] = {"ms_per_step": round(l_ms, 4), "sequences_per_gpu": per_rank,
                                                 "tokens_per_s": round(per_rank * world / (l_ms * 1e-3), 1)}
    return out


def nsplit_8b(H, world, rank, cache=32768):
    """BASELINE configs[4]: the matmuls of an 8B-shape model (hidden 4096, ffn 12288) with packed INT4 group-128
    weights, N-split over the ranks -- rank r runs the HIP kernel on rows row_shard(N, world, r) of the weight
    and its scales, one all_gather_into_tensor (RCCL) reassembles [M, N] on every rank -- and decode attention
    over a 32k-token INT8 cache sharded by kv head with the same gather.  Every rank draws the same full
    weights (one seed), rank 0 also runs the unsplit problem: the N-split result must equal it bit for bit
    (no output element's sum is split), the attention within the attention tolerance (the split count, and so
    the summation order, follows the number of heads a rank holds).  Times are per call, local kernel +
    gather, events on the launching stream, the slowest rank's."""
    import torch.distributed as dist
    from cactus_amd import sharding as S
    comm = S.device_comm()                # the library's own RCCL communicator (cactus_hip_comm_init)
    gen = torch.Generator(device="cuda").manual_seed(4242)       # identical on every rank
    out = {"ranks": world, "matmul_int4": {}, "note": "rows split by cactus_hip_row_shard; cactus_hip_matmul_int4_nsplit_dev "
           "= HIP kernel on the rank's rows + one RCCL exchange on the same stream (M = 1 in place, M = 512 through "
           "rank-major slots and one copy kernel); per call, max over ranks"}
    ok_all = True
    for (N, K) in ((12288, 4096), (4096, 12288), (4096, 4096)):
        B, Bs = make_weights(K, N, True, gen)
        start, end = S.row_shard(N, world, rank)
        Bl, Bsl = B[start:end].contiguous(), Bs[start:end].contiguous()      # the rank's own rows
        if end > start:
            H.pack_weights(Bl)
        for M in (1, 512):
            x = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
            q, sc = H.quantize_activations(x)
            full = torch.empty((M, N), dtype=torch.float16, device="cuda")
            got = comm.matmul_nsplit(q, sc, Bl, Bsl, N, out=full)
            same = True
            if rank == 0:
                same = bool(torch.equal(got.view(torch.int16), H.matmul_int4(q, sc, B, Bs, 128).view(torch.int16)))
            ms = time_events(lambda i: comm.matmul_nsplit(q, sc, Bl, Bsl, N, out=full), 10, warm=2)
            ms_p2p = time_events(lambda i: comm.matmul_nsplit(q, sc, Bl, Bsl, N, out=full, algo=2), 10, warm=2) if M == 1 else 0.0
            ms_local = time_events(lambda i: H.matmul_int4(q, sc, Bl, Bsl, 128), 10, warm=2) if end > start else 0.0
            tt = torch.tensor([ms, ms_local, 0.0 if same else 1.0, ms_p2p], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms, ms_local, bad, ms_p2p = (float(v) for v in tt.tolist())
            ok_all = ok_all and bad == 0.0
            row = {"us": round(ms * 1e3, 2), "us_local_kernel": round(ms_local * 1e3, 2), "bit_identical_to_unsplit": bad == 0.0}
            if M == 1:
                row["us_all_pairs_send_recv"] = round(ms_p2p * 1e3, 2)
            if M == 1:
                row["GBps"] = round(gemv_bytes(M, K, N, True, fused_quant=False) / (ms * 1e-3) / 1e9, 1)
            else:
                row["TOPS"] = round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1)
            out["matmul_int4"][f"M{M}_K{K}_N{N}"] = row
        del B, Bs
    # 32k-context decode attention, 8B heads, sharded by kv head
    Hq, Hkv, D = 32, 8, 128
    if Hkv % world == 0:
        kq = torch.randint(-127, 128, (cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
        vq = torch.randint(-127, 128, (cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen)
        ks = torch.rand((cache, Hkv, 1), device="cuda", generator=gen) * 0.02
        vs = torch.rand((cache, Hkv, 1), device="cuda", generator=gen) * 0.02
        qh = (torch.rand((1, 1, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
        kn = (torch.rand((1, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
        vn = (torch.rand((1, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
        k0, k1, q0, q1 = S.kv_head_shard(Hkv, Hq, world, rank)
        lk, lv = kq[:, k0:k1].contiguous(), vq[:, k0:k1].contiguous()         # the rank's own cache
        lks, lvs = ks[:, k0:k1].contiguous(), vs[:, k0:k1].contiguous()
        lq, lkn, lvn = qh[:, :, q0:q1].contiguous(), kn[:, :, k0:k1].contiguous(), vn[:, :, k0:k1].contiguous()

        gathered = torch.empty((1, Hq * D), dtype=torch.float16, device="cuda")
        mine = gathered[:, q0 * D:q1 * D].view(1, 1, q1 - q0, D)     # the rank's heads: written in place, then exchanged

        def sharded(i=0):
            H.attention_hybrid(lq, lk, lv, lks, lvs, lkn, lvn, 0.0, position_offset=cache, out=mine)
            return comm.allgather_columns(gathered, multiple=(q1 - q0) * D)
        got = sharded()
        diff = 0.0
        if rank == 0:
            full = H.attention_hybrid(qh, kq, vq, ks, vs, kn, vn, 0.0, position_offset=cache)
            diff = float((got.view(-1).float() - full.view(-1).float()).abs().max())
        ms = time_events(sharded, 10, warm=2)
        tt = torch.tensor([ms, diff], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms, diff = (float(v) for v in tt.tolist())
        byts = 2 * cache * Hkv * (D + 4) + 2 * Hkv * D * 2 + 2 * Hq * D * 2
        ok_all = ok_all and diff <= 2e-3
        out["attn_decode_int8kv_c32768_kv_head_shard"] = {"us": round(ms * 1e3, 2), "GBps": round(byts / (ms * 1e-3) / 1e9, 1),
                                                          "max_abs_diff_vs_unsplit": diff}
    out["all_checks_passed"] = ok_all
    comm.close()
    return out


def _time_graph_of(step, reps=5):
    """One call of step() captured into a HIP graph (after one plain call that sizes workspaces), replayed."""
    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def qwen3_decode_step(H, gen, batch, cache_len, layers=28, mixed=False):
    """One decode step of the Qwen3-0.6B op sequence for `batch` sequences that each hold `cache_len`
    cached tokens (cactus_amd/steps.py: Qwen3Decode; embedding lookup to greedy sampler, token ids stay on
    the device).  Synthetic weights, distinct per layer (596 MB of INT8 in total, streamed from HBM every
    step); the whole step is one HIP graph, replayed at fixed positions so every replay does the same work.
    mixed: the converter's MIXED policy (INT4 attn_v / attn_output / MLP in blocks 2..25)."""
    from cactus_amd import steps as ST
    cfg = dict(ST.QWEN3_0P6B, layers=layers)
    w = ST.qwen3_synthetic_weights(gen, cfg, mixed)
    dec = ST.Qwen3Decode(w, cfg, batch, cache_len + 1, cache_gen=gen)
    dec.kv.set_positions([cache_len] * batch)
    dec.kv.tok.copy_(torch.randint(0, cfg["vocab"], (batch,), dtype=torch.int32, device="cuda", generator=gen))
    ms = _time_graph_of(lambda: dec.step(advance=False))
    wbytes = dec.weight_bytes()
    kvbytes = layers * batch * 2 * cache_len * cfg["kv_heads"] * (cfg["head_dim"] + 4)
    return {"ms_per_step": round(ms, 4), "tokens_per_s": round(batch / (ms * 1e-3), 1), "batch": batch,
            "cache_len": cache_len, "GBps": round((wbytes + kvbytes) / (ms * 1e-3) / 1e9, 1),
            "note": "a whole Qwen3-0.6B decode step: embedding lookup to greedy sampler, token ids stay on the device"
                    + ("; MIXED precision policy (INT4 attn_v / attn_output / MLP in blocks 2..25)" if mixed else "")}


def qwen3_decode_via_graph(H, gen, cache_len=512, tokens=24):
    """The same Qwen3-0.6B decode step driven through the graph surface, the way the reference's engine does it
    (cactus_amd/steps.py: Qwen3GraphModel): every token rebuilds the forward pass on a CactusGraph node by node,
    executes it, reads the sampled token back, feeds the KV cache from the k / v nodes and soft-resets.
    Wall-clock per token, host work included; `execute_ms` is the part inside execute() + the token read-back.
    Reported with execute()'s fused launches on and off."""
    import time
    import numpy as np
    from cactus_amd import steps as ST
    cfg = ST.QWEN3_0P6B
    w = ST.qwen3_synthetic_weights(gen, cfg)
    host = {"layers": [{k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v.cpu().numpy())
                        for k, v in L.items()} for L in w["layers"]],
            "emb": tuple(t.cpu().numpy() for t in w["emb"]), "final_norm": w["final_norm"].cpu().numpy()}
    del w
    torch.cuda.empty_cache()
    out = {}
    for fusion in (True, False):
        m = ST.Qwen3GraphModel(host, cfg, max_seq=cache_len + tokens + 8, fusion=fusion)
        rng = np.random.default_rng(3)
        tok = m.step(rng.integers(0, cfg["vocab"], cache_len).tolist())      # prefill fills the cache
        for _ in range(4):
            tok = m.step([tok])
        torch.cuda.synchronize()
        ex_s = 0.0
        orig_execute = m.g.execute

        def timed_execute():
            nonlocal ex_s
            t = time.perf_counter()
            orig_execute()
            torch.cuda.synchronize()
            ex_s += time.perf_counter() - t
        m.g.execute = timed_execute
        t0 = time.perf_counter()
        for _ in range(tokens):
            tok = m.step([tok])
        dt = time.perf_counter() - t0
        out["fused" if fusion else "node_by_node"] = {
            "ms_per_token": round(dt / tokens * 1e3, 3), "tokens_per_s": round(tokens / dt, 1),
            "execute_ms": round(ex_s / tokens * 1e3, 3), "fused_launches_per_token": m.g.fused_launches() // (tokens + 5),
            "pool_allocations": m.g.pool_allocations()}
        m.close()
    out["note"] = ("one sequence, cache_len %d, %d tokens: graph rebuilt per token through the Python binding of the C ABI "
                   "(about 700 node calls), synchronous token read-back" % (cache_len, tokens))
    # the same loop from a C++ host program on the C ABI (examples/qwen_graph_decode.cpp), the model read from a
    # folder in the reference's layout: what the op surface costs a compiled engine
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cactus_amd", "lib", "qwen_graph_decode")
    if os.path.exists(exe):
        import json as _json
        import shutil
        import tempfile
        from cactus_amd import cact
        folder = tempfile.mkdtemp(prefix="cactus_qwen3_")
        try:
            cact.write_qwen_folder(folder, host, cfg)
            prompt = [str(t) for t in np.random.default_rng(3).integers(0, cfg["vocab"], cache_len).tolist()]
            for fusion in (1, 0):
                r = subprocess.run([exe, folder, str(tokens + 2), str(fusion)] + prompt, capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    out["cpp_host_error"] = r.stderr[-300:]
                    break
                d = _json.loads(r.stdout.strip().splitlines()[-1])
                out["cpp_host_fused" if fusion else "cpp_host_node_by_node"] = {
                    "ms_per_token": round(d["ms_per_token"], 3), "tokens_per_s": round(1e3 / d["ms_per_token"], 1),
                    "build_ms": round(d["build_ms"], 3), "execute_ms": round(d["execute_ms"], 3),
                    "cache_update_ms": round(d["cache_update_ms"], 3), "soft_reset_ms": round(d["soft_reset_ms"], 3),
                    "prefill_%d_tokens_ms" % cache_len: round(d["prefill_ms"], 2)}
            # a 4096-token prompt as one chunk through the same surface
            long_prompt = [str(t) for t in np.random.default_rng(4).integers(0, cfg["vocab"], 4096).tolist()]
            r = subprocess.run([exe, folder, "4", "1"] + long_prompt, capture_output=True, text=True, timeout=300)
            if r.returncode == 0:
                d = _json.loads(r.stdout.strip().splitlines()[-1])
                out["cpp_host_prefill_4096"] = {"ms": round(d["prefill_ms"], 2),
                                                "tokens_per_s": round(4096 / (d["prefill_ms"] * 1e-3), 1)}
            # ... and in the engine's default chunks of 256 tokens, a graph per chunk (engine_model.cpp:170-219, 637-642)
            r = subprocess.run([exe, folder, "4", "1"] + long_prompt, capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, CACTUS_PREFILL_CHUNK="256"))
            if r.returncode == 0:
                d = _json.loads(r.stdout.strip().splitlines()[-1])
                out["cpp_host_prefill_4096_chunk256"] = {"ms": round(d["prefill_ms"], 2),
                                                         "tokens_per_s": round(4096 / (d["prefill_ms"] * 1e-3), 1)}
        finally:
            shutil.rmtree(folder, ignore_errors=True)
    return out


def lfm2_decode_step(H, gen, batch, cache_len, mixed=False):
    """One decode step of the LFM2-1.2B op sequence for `batch` independent sequences on one GPU
    (cactus_amd/steps.py: Lfm2Decode; 10 blocks with the short convolution, 6 with attention; dimensions from
    the public model card).  Synthetic INT8 weights (mixed: the converter's MIXED INT4 / INT8 policy, BASELINE
    configs[3]), one HIP graph per step."""
    from cactus_amd import steps as ST
    cfg = ST.LFM2_1P2B
    w = ST.lfm2_synthetic_weights(gen, batch, cfg, mixed)
    dec = ST.Lfm2Decode(w, cfg, batch, cache_len + 1, cache_gen=gen)
    for st in dec.state.values():
        st.copy_((torch.rand(st.shape, device="cuda", generator=gen) - 0.5).half())
    dec.kv.set_positions([cache_len] * batch)
    dec.kv.tok.copy_(torch.randint(0, cfg["vocab"], (batch,), dtype=torch.int32, device="cuda", generator=gen))
    ms = _time_graph_of(lambda: dec.step(advance=False))
    return {"ms_per_step": round(ms, 4), "tokens_per_s": round(batch / (ms * 1e-3), 1), "batch": batch,
            "cache_len": cache_len,
            "note": "a whole LFM2-1.2B-shape decode step (10 conv + 6 attention blocks): embedding lookup to greedy "
                    "sampler, token ids stay on the device; dimensions from the public model card"
                    + ("; MIXED precision policy (INT4 attn_v / attn_output / MLP in blocks 2..13)" if mixed else "")}


def qwen3_prefill(H, gen, tokens=4096, chunk=256, layers=28, mixed=False, lanes=1, lfm2=False):
    """A chunked prefill of one sequence through the Qwen3-0.6B op sequence (cactus_amd/steps.py: Prefill), or with lfm2
    the LFM2-1.2B one (10 short-convolution blocks carrying their rows from chunk to chunk, 6 attention blocks).
    lanes = 1: one HIP graph per chunk, replayed in order (a host that calls cactus_hip_prefill_chunk_dev chunk by chunk);
    lanes > 1: the whole prompt as one graph of cactus_hip_prefill_dev with that many chunks in flight on the library's
    streams.  mixed: the converter's MIXED INT4 / INT8 policy (BASELINE configs[2])."""
    from cactus_amd import steps as ST
    if lfm2:
        cfg = ST.LFM2_1P2B
        w = ST.lfm2_synthetic_weights(gen, 1, cfg, mixed)
    else:
        cfg = dict(ST.QWEN3_0P6B, layers=layers)
        w = ST.qwen3_synthetic_weights(gen, cfg, mixed)
    pf = ST.Prefill(w, cfg, tokens, chunk)
    pf.toks.copy_(torch.randint(0, cfg["vocab"], pf.toks.shape, dtype=torch.int32, device="cuda", generator=gen))
    graphs = []
    if lanes > 1:
        pf.run(lanes)                     # sizes workspaces, builds RoPE rows, creates the streams
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pf.run(lanes)
        graphs.append(g)
    for c in range(pf.nchunks if lanes <= 1 else 0):
        pf.run_chunk(c)                   # sizes workspaces, builds RoPE rows
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            pf.run_chunk(c)
        graphs.append(g)
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 2
    e0.record()
    for _ in range(reps):
        for g in graphs:
            g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"ms": round(ms, 2), "tokens_per_s": round(tokens / (ms * 1e-3), 1), "tokens": tokens, "chunk": chunk,
            "chunks_in_flight": lanes,
            "note": "chunked prefill of one sequence, " + ("LFM2-1.2B" if lfm2 else "Qwen3-0.6B") + " op sequence, synthetic "
                    + ("MIXED INT4 / INT8 weights (INT4 attn_v / attn_output / MLP outside the first and last two blocks)"
                       if mixed else "INT8 weights")}


def extra_table(H, gen, big):
    """Secondary measurements: launches captured in a HIP graph and event timed, rotating weight
    buffers (> 256 MiB in total) so that weights come from HBM."""
    ex = {}
    # the headline step on weights WITHOUT tile-major twins (the reference's row-major [N, K] layout read as it lies)
    sets = [{(K, N): make_weights(K, N, False, gen) for K in SIZES for N in SIZES} for _ in range(4)]
    acts = {K: (torch.rand((1, K), device="cuda", generator=gen) * 2 - 1).half() for K in SIZES}
    outs = {(K, N): torch.empty((1, N), dtype=torch.float16, device="cuda") for K in SIZES for N in SIZES}
    order = sorted(((K, N) for K in SIZES for N in SIZES), key=lambda kn: -kn[0] * kn[1])
    batches = [H.MatmulBatch([(acts[K], ws[(K, N)][0], ws[(K, N)][1], outs[(K, N)]) for K, N in order], 128) for ws in sets]
    ms = time_graph(lambda i: batches[i & 3].run(), 12)
    sb = sum(gemv_bytes(1, K, N) for K in SIZES for N in SIZES)
    ex["gemv_sweep_row_major_layout"] = {"us_per_step": round(ms * 1e3, 2), "GBps": round(sb / (ms * 1e-3) / 1e9, 1),
                                         "frac_hbm": round(sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                         "note": "the same batched launch reading the row-major weights (no twins)"}
    del sets, batches
    torch.cuda.empty_cache()
    # per-shape GEMV, INT8 and packed INT4, M = 1 and M = 16
    for int4 in (False, True):
        tag = "int4" if int4 else "int8"
        for M in (1, 16):
            rows = {}
            for K, N in ((1024, 1024), (4096, 4096), (8192, 8192), (1024, 151936)):
                nb = max(2, int(600e6 // max(1, (N * K // (2 if int4 else 1)))) + 1)
                nb = min(nb, 64)
                ws = [make_weights(K, N, int4, gen, pack=True) for _ in range(nb)]
                x = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
                o = torch.empty((M, N), dtype=torch.float16, device="cuda")
                ms = time_graph(lambda i: H.matmul_f16act(x, ws[i % nb][0], ws[i % nb][1], 128, int4=int4, out=o),
                                max(nb, 16))
                gb = gemv_bytes(M, K, N, int4) / (ms * 1e-3) / 1e9
                rows[f"K{K}_N{N}"] = {"us": round(ms * 1e3, 2), "GBps": round(gb, 1),
                                      "frac_hbm": round(gb / HBM_PEAK_GBPS, 3)}
                if M > 4:     # rows the launch does not quantize itself: also on INT8 rows, the reference kernel's own signature
                    q, sc = H.quantize_activations(x)
                    fn = H.matmul_int4 if int4 else H.matmul_int8
                    ms = time_graph(lambda i: fn(q, sc, ws[i % nb][0], ws[i % nb][1], 128, out=o), max(nb, 16))
                    gb = gemv_bytes(M, K, N, int4) / (ms * 1e-3) / 1e9
                    rows[f"K{K}_N{N}"].update({"us_int8_rows_in": round(ms * 1e3, 2), "frac_hbm_int8_rows_in": round(gb / HBM_PEAK_GBPS, 3)})
                del ws            # (a packed tensor's twin goes with it)
            ex[f"gemv_{tag}_M{M}"] = rows
    # M = 512 GEMM: MFMA rate
    for int4 in (False, True):
        rows = {}
        for K, N in ((4096, 4096), (8192, 8192)):
            B, Bs = make_weights(K, N, int4, gen)
            x = (torch.rand((512, K), device="cuda", generator=gen) * 2 - 1).half()
            q, s = H.quantize_activations(x)
            o = torch.empty((512, N), dtype=torch.float16, device="cuda")
            fn = H.matmul_int4 if int4 else H.matmul_int8
            ms = time_events(lambda i: fn(q, s, B, Bs, 128, out=o), 20, warm=3)
            tops = 2.0 * 512 * N * K / (ms * 1e-3) / 1e12
            rows[f"K{K}_N{N}"] = {"us": round(ms * 1e3, 1), "TOPS": round(tops, 1),
                                  "frac_int8_mfma_peak": round(tops / INT8_MFMA_PEAK_TOPS, 4)}
        ex["gemm_%s_M512" % ("int4" if int4 else "int8")] = rows
    # FP16 x FP16^T (cactus_matmul_f16), fp32 accumulation on the matrix cores
    rows = {}
    for M, K, N in ((512, 4096, 4096), (512, 8192, 8192), (1, 4096, 4096)):
        a16 = (torch.rand((M, K), device="cuda", generator=gen) * 2 - 1).half()
        b16 = (torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).half()
        o = torch.empty((M, N), dtype=torch.float16, device="cuda")
        ms = time_events(lambda i: H.matmul_f16(a16, b16, out=o), 20, warm=3)
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        rows[f"M{M}_K{K}_N{N}"] = {"us": round(ms * 1e3, 1), "TFLOPs": round(tf, 1),
                                   "frac_fp16_mfma_peak": round(tf / 2500.0, 4)}
        if M == 1:      # a GEMV: HBM-bound; 8 rotating weights, row-major and with tile-major twins
            bs = [b16] + [(torch.rand((N, K), device="cuda", generator=gen) * 2 - 1).half() for _ in range(7)]
            for tag in ("row_major", "twins"):
                if tag == "twins":
                    for b_ in bs:
                        H.pack_weights(b_)
                ms = time_graph(lambda i: H.matmul_f16(a16, bs[i % 8], out=o), 16)
                gb = (2.0 * N * K + 2 * K + 2 * N) / (ms * 1e-3) / 1e9
                rows[f"M{M}_K{K}_N{N}_gemv_{tag}"] = {"us": round(ms * 1e3, 2), "GBps": round(gb, 1),
                                                      "frac_hbm": round(gb / HBM_PEAK_GBPS, 3)}
            del bs
        del a16, b16
    ex["matmul_f16"] = rows
    # decode attention over the INT8 KV cache
    rows = {}
    for name, Hq, Hkv, D, cache, Bn in (("qwen3_c512", 16, 8, 128, 512, 1), ("qwen3_c4096", 16, 8, 128, 4096, 1),
                                        ("8b_c32768", 32, 8, 128, 32768, 1), ("qwen3_c4096_b64", 16, 8, 128, 4096, 64),
                                        ("qwen3_c512_b256", 16, 8, 128, 512, 256), ("lfm2_c512_b1024", 32, 8, 64, 512, 1024),
                                        ("gemma270m_c512_b1024", 4, 1, 256, 512, 1024)):
        ng = (D + 127) // 128      # scale groups per cached row (kernel.h:18)
        # cache sets used in turn, more than 600 MB in all (at least 2): a set must not still sit in the 256 MB
        # Infinity Cache when its turn comes again, as a layer's cache does not in a decode step (the same 280 MB cache
        # every launch reads 15 % faster than HBM delivers)
        set_bytes = 2 * Bn * cache * Hkv * (D + 4 * ng)
        nsets = max(2, min(16, int(600e6 // set_bytes) + 1))
        sets = []
        for _ in range(nsets):
            sets.append((torch.randint(-127, 128, (Bn, cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen),
                         torch.randint(-127, 128, (Bn, cache, Hkv, D), dtype=torch.int8, device="cuda", generator=gen),
                         torch.rand((Bn, cache, Hkv, ng), device="cuda", generator=gen) * 0.02,
                         torch.rand((Bn, cache, Hkv, ng), device="cuda", generator=gen) * 0.02))
        qh = (torch.rand((Bn, 1, Hq, D), device="cuda", generator=gen) * 2 - 1).half()
        kn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
        vn = (torch.rand((Bn, 1, Hkv, D), device="cuda", generator=gen) * 2 - 1).half()
        lens = torch.full((Bn,), cache, dtype=torch.int32, device="cuda")
        out = torch.empty_like(qh)
        ms = time_graph(lambda i: H.attention_hybrid_batched(qh, *sets[i % nsets], kn, vn, 0.0, lens, lens, out=out),
                        max(8, 2 * nsets))
        byts = Bn * (2 * cache * Hkv * (D + 4 * ng) + 2 * Hkv * D * 2 + 2 * Hq * D * 2)
        gb = byts / (ms * 1e-3) / 1e9
        rows[name] = {"us": round(ms * 1e3, 2), "GBps": round(gb, 1), "frac_hbm": round(gb / HBM_PEAK_GBPS, 3),
                      "cache_sets": nsets}
        del sets
        torch.cuda.empty_cache()
    ex["attn_decode_int8kv"] = rows
    # the ops between the hot kernels of a decoder block, prefill-sized (4096 tokens of Qwen3-0.6B):
    # HBM-bound; bytes = fp16 in + out (+ second input)
    rows = {}
    half = lambda *sh: (torch.rand(sh, device="cuda", generator=gen) * 2 - 1).half()
    T, Hd, Hq, Hkv, D, F = 4096, 1024, 16, 8, 128, 3072
    xa, xb, wn, o1, o2 = half(T, Hd), half(T, Hd), half(Hd), half(T, Hd), half(T, Hd)
    q4, k4, qw, kw = half(1, T, Hq, D), half(1, T, Hkv, D), half(D), half(D)
    oq, ok = torch.empty_like(q4), torch.empty_like(k4)
    ga, up, oa = half(T, F), half(T, F), half(T, F)
    H.rope_reserve(D, 1e6, T)
    for name, fn, nbytes in (
            ("rms_norm_4096x1024", lambda i: H.rms_norm(xa, wn, 1e-6, out=o1), 4 * T * Hd),
            ("add_rms_norm_4096x1024", lambda i: H.add_rms_norm(xa, xb, wn, 1e-6, out_sum=o1, out_norm=o2), 8 * T * Hd),
            ("qk_norm_rope_4096x(16+8)x128", lambda i: H.qk_norm_rope(q4, k4, qw, kw, 1e-6, 1e6, 0, out_q=oq, out_k=ok),
             4 * T * (Hq + Hkv) * D),
            ("silu_multiply_4096x3072", lambda i: H.silu_multiply(ga, up, out=oa), 6 * T * F)):
        ms = time_graph(fn, 8)
        gb = nbytes / (ms * 1e-3) / 1e9
        rows[name] = {"us": round(ms * 1e3, 2), "GBps": round(gb, 1), "frac_hbm": round(gb / HBM_PEAK_GBPS, 3)}
    ex["block_ops_prefill"] = rows
    del xa, xb, o1, o2, q4, k4, oq, ok, ga, up, oa
    # LFM2-1.2B short-convolution block (hidden 2048, 3 taps, 2 carried rows): bytes = the
    # in-projection's B, C-gate, X rows in + gated rows out + the state read and written
    rows = {}
    Cc, Kc, Wc = 2048, 3, 2
    wconv = half(Cc, Kc)
    for name, nseq, Lc in (("decode_b1024", 1024, 1), ("decode_b128", 128, 1), ("prefill_1x4096", 1, 4096),
                           ("prefill_16x256", 16, 256)):
        ipj, stt = half(nseq, Lc, 3 * Cc), torch.zeros((nseq, Wc, Cc), dtype=torch.float16, device="cuda")
        og = torch.empty((nseq, Lc, Cc), dtype=torch.float16, device="cuda")
        ms = time_graph(lambda i: H.lfm2_conv_gate(ipj, wconv, stt, out=og), 8)
        nbytes = nseq * Cc * (8 * Lc + 4 * Wc)
        gb = nbytes / (ms * 1e-3) / 1e9
        rows[name] = {"us": round(ms * 1e3, 2), "GBps": round(gb, 1), "frac_hbm": round(gb / HBM_PEAK_GBPS, 3)}
    ex["lfm2_conv_block"] = rows
    del ipj, stt, og
    torch.cuda.empty_cache()
    # FP16 causal prefill attention (Qwen3 heads), one layer: the matrix-core kernel
    rows = {}
    for S in (256, 1024, 4096):
        q = (torch.rand((1, S, 16, 128), device="cuda", generator=gen) * 2 - 1).half()
        k = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
        v = (torch.rand((1, S, 8, 128), device="cuda", generator=gen) * 2 - 1).half()
        o = torch.empty_like(q)
        ms = time_graph(lambda i: H.attention_f16(q, k, v, 0.0, out=o), 4)
        fl = 4.0 * S * S / 2 * 16 * 128        # causal: half of the S x S scores
        rows[f"S{S}"] = {"us": round(ms * 1e3, 1), "TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1),
                         "frac_fp16_mfma_peak": round(fl / (ms * 1e-3) / 2.5e15, 4)}
    ex["attn_prefill_f16_causal"] = rows
    # a 256-token prefill chunk against the INT8 cache (the chunks after the first)
    rows = {}
    for cache in (256, 2048, 3840):
        S, Hq_, Hkv_, D_ = 256, 16, 8, 128
        q = (torch.rand((1, S, Hq_, D_), device="cuda", generator=gen) * 2 - 1).half()
        kn = (torch.rand((1, S, Hkv_, D_), device="cuda", generator=gen) * 2 - 1).half()
        vn = (torch.rand((1, S, Hkv_, D_), device="cuda", generator=gen) * 2 - 1).half()
        kc = torch.randint(-127, 128, (cache, Hkv_, D_), dtype=torch.int8, device="cuda", generator=gen)
        vc = torch.randint(-127, 128, (cache, Hkv_, D_), dtype=torch.int8, device="cuda", generator=gen)
        ks = torch.rand((cache, Hkv_, 1), device="cuda", generator=gen) * 0.02
        vs = torch.rand((cache, Hkv_, 1), device="cuda", generator=gen) * 0.02
        H.attention_hybrid(q, kc, vc, ks, vs, kn, vn, 0.0, position_offset=cache)      # sizes the workspace
        ms = time_graph(lambda i: H.attention_hybrid(q, kc, vc, ks, vs, kn, vn, 0.0, position_offset=cache), 4)
        fl = 4.0 * S * (cache + S / 2) * Hq_ * D_
        rows[f"cache{cache}"] = {"us": round(ms * 1e3, 1), "TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1)}
    ex["attn_prefill_chunk256_int8kv"] = rows
    # decode tail at the Qwen3 vocabulary: the stochastic sampler (engine defaults: temperature
    # 0.6, top-p 0.95, top-k 20) against the greedy one, and the entropy figure
    rows = {}
    for batch in (1, 256):
        lg = (torch.randn((batch, 151936), device="cuda", generator=gen) * 3).half()
        dr = torch.rand((batch,), device="cuda", generator=gen)
        smp = H.Sampler(batch, 151936)
        tok = torch.empty((batch,), dtype=torch.int32, device="cuda")
        ent = torch.empty((batch,), dtype=torch.float32, device="cuda")
        ms_s = time_graph(lambda i: smp.sample(lg, 0.6, 0.95, 20, dr, out=tok, check=False), 8)
        ms_g = time_graph(lambda i: H.sample_greedy(lg, out=tok), 8)
        ms_e = time_graph(lambda i: H.logits_entropy(lg, out=ent), 8)
        rows[f"b{batch}_v151936"] = {"sample_topk20_topp_us": round(ms_s * 1e3, 2), "greedy_us": round(ms_g * 1e3, 2),
                                     "entropy_us": round(ms_e * 1e3, 2)}
        del lg, smp
    ex["decode_tail"] = rows
    # Model::prefill's chunk_size (engine_model.cpp:170; 256 unless the backend names its own, :637-642): the
    # engine's default, and chunks sized for this device (one chunk: FP16 attention over the whole prompt)
    ex["qwen3_0p6b_prefill"] = {"t4096_chunk256": qwen3_prefill(H, gen, 4096, 256),
                                "t4096_chunk1024": qwen3_prefill(H, gen, 4096, 1024),
                                "t4096_chunk4096": qwen3_prefill(H, gen, 4096, 4096),
                                "t4096_chunk256_4_in_flight": qwen3_prefill(H, gen, 4096, 256, lanes=4),
                                "t4096_chunk1024_4_in_flight": qwen3_prefill(H, gen, 4096, 1024, lanes=4),
                                "t4096_chunk256_mixed_int4": qwen3_prefill(H, gen, 4096, 256, mixed=True),
                                "t4096_chunk256_mixed_int4_4_in_flight": qwen3_prefill(H, gen, 4096, 256, mixed=True, lanes=4),
                                "t4096_chunk4096_mixed_int4": qwen3_prefill(H, gen, 4096, 4096, mixed=True)}
    torch.cuda.empty_cache()
    ex["qwen3_0p6b_decode_step"] = {"b1_c512": qwen3_decode_step(H, gen, 1, 512),
                                       "b256_c512": qwen3_decode_step(H, gen, 256, 512),
                                       "b256_c512_mixed_int4": qwen3_decode_step(H, gen, 256, 512, mixed=True)}
    torch.cuda.empty_cache()
    ex["qwen3_decode_step_via_graph"] = qwen3_decode_via_graph(H, gen, 512)
    torch.cuda.empty_cache()
    # config 4 (1024 independent LFM2-1.2B sequences sharded over the GPUs of a node): what one GPU
    # runs at 8 GPUs (128 sequences) and at 1 GPU (1024)
    ex["lfm2_1p2b_prefill"] = {"t4096_chunk256": qwen3_prefill(H, gen, 4096, 256, lfm2=True),
                               "t4096_chunk256_4_in_flight": qwen3_prefill(H, gen, 4096, 256, lfm2=True, lanes=4),
                               "t4096_chunk256_mixed_int4_4_in_flight": qwen3_prefill(H, gen, 4096, 256, lfm2=True, mixed=True, lanes=4),
                               "t4096_chunk4096": qwen3_prefill(H, gen, 4096, 4096, lfm2=True)}
    ex["lfm2_1p2b_decode_step"] = {"b128_c512": lfm2_decode_step(H, gen, 128, 512),
                                      "b1024_c512": lfm2_decode_step(H, gen, 1024, 512),
                                      "b1024_c512_mixed_int4": lfm2_decode_step(H, gen, 1024, 512, mixed=True)}
    return ex


if __name__ == "__main__":
    main()
