#!/usr/bin/env python3
"""Headline benchmark: ListMLE training step of the D-MPNN reaction scorer on MI355X.

Workload (BASELINE.json configs[2]): ListMLE over queries of 64 candidates, model
build_model(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True,
dropout=0.1, task_num=1, add_features_dim=1) in train mode; a "step" is one optimizer step over
`--queries-per-step` whole queries (default 64 -> 4096 candidates, ~73k atoms / ~139k directed
bonds per side): forward (encoder on reactants and products, diff encoder, FFN) + ListMLE loss +
backward + gradient all-reduce (N > 1) + Adam/NoamLR update.  Graphs are pre-packed and resident
in HBM before the timed region; a pool of distinct steps is cycled so no step's activations stay
in the 256 MiB Infinity Cache between uses.  The 100k-query epoch is 1563 such steps; `--steps`
of them are timed.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement): value = whole-job queries/s.
Extra objects: `roofline` (dominant kernel, live HIP-event timing over the timed region),
`roofline_gather` (the gather kernel vs HBM), `cpu_baseline` (the CPU oracle on this box's host
cores, rank 0 / N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def build_pool(args, rank, device):
    from reactranker_amd import featurization, synth
    pool = []
    t_gen = t_pack = 0.0
    for i in range(args.pool):
        t0 = time.time()
        qb = synth.make_queries(1000 * (rank + 1) + i, args.queries_per_step, args.cands)
        t1 = time.time()
        rb = featurization.BatchMolGraph(qb.r_specs, K=args.pad_width)      # global pad width (hazard H1)
        pb = featurization.BatchMolGraph(qb.p_specs, K=args.pad_width)
        rg, pg = rb.device_graph(device), pb.device_graph(device)
        ub, _, _ = rb.unique()                                # distinct reactants (used when dropout is inactive)
        ub.device_graph(device)
        t2 = time.time()
        t_gen += t1 - t0
        t_pack += t2 - t1
        pool.append(dict(r=rb, p=pb, scope=qb.scope, targets=torch.tensor(qb.targets).to(device),
                         add=torch.tensor(qb.add_features).to(device), qb=qb if i == 0 else None))
    return pool, t_gen, t_pack


def cpu_baseline(args, qb0):
    """The CPU oracle (vectorised PyTorch-CPU restatement, pinned to the reference by golden vectors)
    on this box's host cores: fwd + ListMLE + bwd + Adam over a bounded sample of the same workload."""
    from oracle import ref_cpu as O
    from reactranker_amd import synth
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = max(1, min(cores, args.cpu_threads))          # a 1-GPU box owns a 16-core share of the host
    torch.set_num_threads(cores)
    shapes = O.model_shapes(args.hidden, args.depth, args.depth, 3, 1, 1, True)
    P = O.params_from_numpy(synth.seeded_weights(shapes, 0), requires_grad=True)
    params = [p for p in P.values() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    cfg = dict(depth=args.depth, diff_depth=args.depth, ffn_depth=3, task_type="with_softplus")
    rg, pg = O.graph_tensors(O.pack_batch(qb0.r_specs, K=args.pad_width)), \
        O.graph_tensors(O.pack_batch(qb0.p_specs, K=args.pad_width))
    tt = torch.tensor(qb0.targets)

    def step(faithful, nq):
        m = sum(qb0.scope[:nq])
        if nq < len(qb0.scope):
            rs, ps = qb0.r_specs[:m], qb0.p_specs[:m]
            r, p = O.graph_tensors(O.pack_batch(rs, K=args.pad_width)), O.graph_tensors(O.pack_batch(ps, K=args.pad_width))
        else:
            r, p = rg, pg
        t0 = time.time()
        out = O.reaction_forward(P, cfg, r, p, qb0.add_features[:m], faithful=faithful)
        loss = O.listmle_loss(out, qb0.scope[:nq], tt[:m])
        opt.zero_grad()
        loss.sum().backward()
        opt.step()
        return time.time() - t0

    nq = len(qb0.scope)
    step(False, min(nq, 8))                     # warm the allocator / thread pool
    reps, spent = 0, 0.0
    while reps < 8 and spent < 15.0:
        spent += step(False, nq)
        reps += 1
        log(f"  cpu step {reps}: cumulative {spent:.1f}s")
    vec = dict(value=round(reps * nq / spent, 3), unit="queries/s", cores=cores, kind="port",
               sample=f"{reps} steps x {nq} queries x {args.cands} candidates, fwd+ListMLE+bwd+Adam, "
                      f"vectorised CPU oracle (oracle/ref_cpu.py), torch {torch.__version__} CPU, {cores} threads")
    nqf = min(nq, 16)
    tf = step(True, nqf)
    vec["faithful_variant"] = dict(value=round(nqf / tf, 3), unit="queries/s",
                                   sample=f"1 step x {nqf} queries: keeps the reference's per-molecule readout loop "
                                          f"(models/mpn.py:224-235, backward quadratic in batch size)")
    return vec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queries-per-step", type=int, default=64)
    ap.add_argument("--cands", type=int, default=64)
    ap.add_argument("--pool", type=int, default=6, help="distinct pre-packed steps cycled through")
    ap.add_argument("--hidden", type=int, default=300)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--pad-width", type=int, default=4, help="global a2b pad width K (same on every rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads for the CPU baseline (box share = 16)")
    ap.add_argument("--no-profile", action="store_true", help="skip the live HIP-event kernel timing")
    ap.add_argument("--fwd-only", action="store_true", help="(default) also time forward+loss alone, reported as extra")
    ap.add_argument("--no-fwd-only", action="store_true", help="skip the forward+loss-only measurement")
    ap.add_argument("--no-side-stream", action="store_true", help="run weight-gradient GEMMs on the main stream")
    ap.add_argument("--no-aux-stream", action="store_true", help="run the reactant encoder on the main stream")
    ap.add_argument("--aux-backward", action="store_true", help="also run the reactant encoder's backward on the aux stream")
    ap.add_argument("--foreach-adam", action="store_true", help="torch's multi-kernel Adam instead of its fused single-kernel one")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("RR_SINGLE_DEVICE"):                # rehearsal of the N>1 path on a one-GPU box
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm (xGMI underneath); RR_DIST_BACKEND=gloo only for one-GPU rehearsals
        dist.init_process_group(os.environ.get("RR_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    from reactranker_amd import functions as Fn
    from reactranker_amd import loss as RL
    from reactranker_amd.base_model import build_model
    from reactranker_amd.dp import GradBucket

    Fn.SideStream.enabled = not args.no_side_stream
    Fn.AuxStream.enabled = not args.no_aux_stream
    if args.aux_backward:
        Fn.AuxStream.backward = True
    torch.manual_seed(0)                                  # identical replicas
    model = build_model(hidden_size=args.hidden, mpnn_depth=args.depth, mpnn_diff_depth=args.depth, ffn_depth=3,
                        use_bias=True, dropout=args.dropout, task_num=1, ffn_last_layer="with_softplus",
                        add_features_dim=1).to(device)
    model.train()
    # the reference's build_optimizer / NoamLR (train/utils.py) through their mirrors; fused Adam = same update, one kernel
    from reactranker_amd.train_utils import build_lr_scheduler, build_optimizer
    opt = build_optimizer(model, fused=not args.foreach_adam)
    bucket = GradBucket(model.parameters())
    mle = RL.MLEloss()
    log("building the step pool (synthetic graphs -> native packer -> HBM)")
    pool, t_gen, t_pack = build_pool(args, rank, device)
    log(f"pool ready: {len(pool)} steps, gen {t_gen:.1f}s pack+upload {t_pack:.1f}s; warmup")
    # 100k queries / 64 per step = 1562 steps per epoch; 2 warm-up epochs of 25 (main.py defaults: 1e-4 -> 1e-3 -> 1e-4)
    sched = build_lr_scheduler(opt, warmup_epochs=2, total_epochs=25, train_data_size=100000,
                               batch_size=args.queries_per_step, init_lr=1e-4, max_lr=1e-3, final_lr=1e-4)

    state = dict(step=0)

    def train_step(i):
        b = pool[i % len(pool)]
        out = model(b["r"], b["p"], gpu=local, add_features=b["add"])
        loss = mle(out, b["scope"], b["targets"], local)
        opt.zero_grad(set_to_none=True)
        loss.sum().backward()
        bucket.allreduce(1.0 / world)                     # equal shards: mean of per-rank query-mean grads
        state["step"] += 1
        sched.step()                                      # NoamLR writes param_groups[0]['lr'] (train/utils.py:88)
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the warm-up steps are timed per launch for every heavy kernel, to find the dominant MFMA kernel; the timed
    # region then carries events around that kernel and the gather kernel only (an event pair around all ~60
    # heavy launches of a step costs ~4 % of the step)
    dominant = None
    if not args.no_profile:
        Fn.Profiler.start()
    for i in range(args.warmup):
        train_step(i)
    fence()
    if not args.no_profile:
        tot = {}
        for key, flops, _, e0, e1 in Fn.Profiler.stop():
            if flops:
                tot[key] = tot.get(key, 0.0) + e0.elapsed_time(e1)
        dominant = max(tot, key=tot.get) if tot else None
    log(f"timing {args.steps} steps (live events on: {dominant}, gather_sum_kernel)")
    if not args.no_profile:
        Fn.Profiler.start(only=[k for k in (dominant, "gather_sum_kernel") if k])
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        if not args.no_profile:                           # events around the launches of every 5th step only: an event
            Fn.Profiler.enabled = (i % 5 == 0)            # pair costs ~14 us of stream time (5 % of the step if always on)
        last = train_step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    records = Fn.Profiler.stop() if not args.no_profile else []
    log(f"timed region done: {elapsed / max(1, args.steps) * 1e3:.2f} ms/step")
    loss_val = float(last.detach().cpu()) if last is not None else float("nan")
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    extra = {}
    if not args.no_fwd_only:                              # report fwd+loss and fwd+loss+bwd separately
        model.eval()
        with torch.no_grad():                             # upload the de-duplication maps outside the timed loop
            for b in pool:
                model(b["r"], b["p"], gpu=local, add_features=b["add"])
        fence()
        tf0 = time.perf_counter()
        with torch.no_grad():
            for i in range(args.steps):
                b = pool[i % len(pool)]
                mle(model(b["r"], b["p"], gpu=local, add_features=b["add"]), b["scope"], b["targets"], local)
        fence()
        extra["fwd_loss_queries_per_s"] = round(world * args.steps * args.queries_per_step / (time.perf_counter() - tf0), 1)
        extra["fwd_loss_note"] = "eval mode (no dropout): forward + ListMLE only; reactant encoder runs once per distinct reactant"
        model.train()

    # HBM traffic per launch comes from separate rocprofv3 --pmc passes (tools/traffic_from_pmc.py)
    traffic = {}
    try:
        import glob
        tf = sorted(glob.glob(os.path.join(REPO, "profiles", "*_traffic.json")))
        if tf:
            traffic = {k: v["hbm_bytes_per_launch"] for k, v in json.load(open(tf[-1]))["kernels"].items()}
    except Exception:
        traffic = {}

    def tr(key):
        k = key.replace("gather_sum_kernel", "gather_sum_kernel<4>")
        return traffic.get(k)

    # ---- per-kernel live timings (HIP events on the launch stream)
    def summarise(recs):
        roof, roof_g, ktable = None, None, {}
        if not recs:
            return roof, roof_g, ktable
        agg = {}
        for key, flops, nbytes, e0, e1 in recs:
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += flops
            a[3] += nbytes
        for k, (n, secs, fl, by) in agg.items():
            ktable[k] = dict(launches=n, avg_us=round(secs / n * 1e6, 2), total_ms=round(secs * 1e3, 3),
                             tflops=round(fl / secs / 1e12, 2) if fl else None,
                             algo_gbs=round(by / secs / 1e9, 1))
        mf = {k: v for k, v in agg.items() if v[2] > 0}
        if mf:
            dom = max(mf, key=lambda k: mf[k][1])
            n, secs, fl, by = mf[dom]
            ach = fl / secs / 1e12
            roof = dict(kernel=dom, bound="mfma", achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), traffic=tr(dom), launches=n,
                        avg_launch_us=round(secs / n * 1e6, 2),
                        algorithmic_flops_per_launch=round(fl / n), algorithmic_bytes_per_launch=round(by / n))
        if "gather_sum_kernel" in agg:
            n, secs, fl, by = agg["gather_sum_kernel"]
            ach = by / secs / 1e9
            roof_g = dict(kernel="gather_sum_kernel", bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS,
                          unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4), traffic=tr("gather_sum_kernel"), launches=n,
                          avg_launch_us=round(secs / n * 1e6, 2), algorithmic_bytes_per_launch=round(by / n))
        return roof, roof_g, ktable

    # timed region: kernels of the three streams overlap, so a kernel's launch duration includes the time it
    # shares the chip with kernels of the other streams (this is what rocprofv3 --stats of this command shows)
    roof, roof_g, ktable = summarise(records)
    load_clock = {"clock_ghz": 1.8, "peak": round(PEAK_F32_MFMA_TFLOPS * 1.8 / 2.4, 1), "unit": "TFLOP/s",
                  "source": "shader clock measured inside the k-loop of this kernel under load (s_memtime / s_memrealtime, "
                            "tools/trace_linear.py, profiles/r01_linear_phase_trace.txt); the datasheet peak assumes 2.4 GHz"}
    if roof:
        roof["peak_at_load_clock"] = load_clock
    if roof:
        roof["note"] = "timed region; weight-gradient / reactant-encoder streams run concurrently with the main stream"
    # isolated pass: the same steps with every kernel serialised on one stream -> per-kernel quality
    roof_iso = roof_g_iso = None
    ktable_iso = {}
    if records:
        side0, aux0 = Fn.SideStream.enabled, Fn.AuxStream.enabled
        Fn.SideStream.enabled = Fn.AuxStream.enabled = False
        fence()
        Fn.Profiler.start()
        for i in range(min(args.steps, 10)):
            train_step(args.warmup + args.steps + i)
        fence()
        roof_iso, roof_g_iso, ktable_iso = summarise(Fn.Profiler.stop())
        Fn.SideStream.enabled, Fn.AuxStream.enabled = side0, aux0
        if roof_iso:
            roof_iso["peak_at_load_clock"] = load_clock
            roof_iso["note"] = "extra pass after the timed region, one stream (kernels do not overlap)"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("CPU baseline (oracle) on the host cores")
        cpu = cpu_baseline(args, pool[0]["qb"])
        log("CPU baseline done")

    if rank == 0:
        qps = world * args.steps * args.queries_per_step / elapsed
        g0 = pool[0]["p"].device_graph(device)
        line = {
            "metric": "queries/sec (lists scored+loss) ListMLE", "value": round(qps, 2), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ListMLE 100k queries x 64 candidates (BASELINE.json configs[2]), "
                                   f"{args.queries_per_step}-query steps, D-MPNN depth={args.depth} hidden={args.hidden}",
                       "phase": "train step: fwd + ListMLE + bwd + grad all-reduce + Adam/NoamLR",
                       "queries_per_step_per_gpu": args.queries_per_step, "candidates_per_query": args.cands,
                       "atoms_per_step_side": int(g0.nA), "directed_bonds_per_step_side": int(g0.nB),
                       "pad_width_K": int(g0.K), "dropout": args.dropout, "step_pool": len(pool),
                       "parallelism": f"dp{world} (whole queries per rank, one RCCL all-reduce of the flat fp32 "
                                      f"gradient bucket per step)"},
            "roofline": roof, "roofline_gather": roof_g, "roofline_isolated": roof_iso,
            "roofline_gather_isolated": roof_g_iso, "cpu_baseline": cpu,
            "kernels": ktable, "kernels_isolated": ktable_iso, "final_loss": round(loss_val, 6),
            "host_prep_s": {"synthetic_generation": round(t_gen, 2), "native_pack_and_upload": round(t_pack, 2)},
        }
        line.update(extra)
        if cpu:
            line["gpu_over_cpu"] = round(qps / cpu["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
