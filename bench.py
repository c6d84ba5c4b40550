#!/usr/bin/env python3
"""Headline benchmark: training step of the D-MPNN reaction scorer + ranking loss on MI355X.

Headline workload (BASELINE.json configs[2], preset `mle64`): ListMLE over queries of 64 candidates, model
build_model(hidden_size=300, mpnn_depth=3, mpnn_diff_depth=3, ffn_depth=3, use_bias=True, dropout=0.1, task_num=1,
add_features_dim=1) in train mode; a "step" is one optimizer step over 64 whole queries (4096 candidates, ~71k atoms /
~139k directed bonds per side): forward (encoder on reactants and products, diff encoder, FFN) + ListMLE + backward +
gradient all-reduce (N > 1) + Adam/NoamLR.  Graphs are pre-packed and resident in HBM before the timed region; a pool
of distinct steps is cycled so no step's activations stay in the 256 MiB Infinity Cache between uses.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python bench.py --gpus N ...          (N > 1 without a launcher: starts `python -m torch.distributed.run` itself as a child)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line of at most 6 KB (contract in the task statement; tests/test_bench_line_cpu.py holds the size and
the keys): value = whole-job queries/s of the headline preset in the library's DEFAULT arithmetic - f32 in / out /
accumulate, every encoder multiply through three exact bf16 terms per operand (no operand bit dropped).  In the line:
  roofline / roofline_gather   dominant MFMA kernel and the gather kernel, live HIP-event timing in situ
  cpu_baseline                 the CPU oracle on this box's host cores (rank 0 / N=1 only, bounded sample)
  f32_mfma_path, f16x2_path    the same steps with every GEMM on the f32 MFMA / on the opt-in two-f16-term form (own dtype string)
  epoch_stream                 the same training step fed from shard files on disk over >= 200 DISTINCT steps
  presets                      the other BASELINE configurations at full step size: queries/s and ms/step each
  dp                           (N > 1) the ranks RCCL saw and the all-reduce's duration from HIP events
Everything else - per-kernel tables, isolated rooflines, the eval path's roofline, notes - goes to bench_detail.json (next to
this file and, when it exists, gpurun_out/) and to stderr.
"""
import argparse
import gc
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
SPLIT_PRODUCTS = 6                # bf16 MFMA products issued per f32 multiply on the three-term split path
F16X2_PRODUCTS = 3                # f16 MFMA products per multiply on the two-term path (same 2.5 PF dense peak)
DTYPE_BF16X3 = "f32"              # f32 in / out / accumulate; every operand bit enters the products (three exact bf16 terms)
DTYPE_F16X2 = "f32 io, 2xf16 22-bit operands"   # the opt-in two-term form rounds operands to 22 bits: not an f32 claim
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)

PRESETS = {
    "mle64": dict(loss="mle", cands=64, queries=64, hidden=300, depth=3, task_num=1, last="with_softplus", task_type=None,
                  dropout=0.1, metric="queries/sec (lists scored+loss) ListMLE",
                  workload="ListMLE 100k queries x 64 candidates (BASELINE.json configs[2])"),
    "listnet32": dict(loss="listnet", cands=32, queries=64, hidden=300, depth=3, task_num=1, last="with_softplus",
                      task_type=None, dropout=0.1, metric="queries/sec (lists scored+loss) ListNet",
                      workload="ListNet 10k queries x 32 candidates (BASELINE.json configs[1])"),
    "ranknet64": dict(loss="ranknet", cands=64, queries=256, hidden=300, depth=3, task_num=1, last="no_softplus",
                      task_type=None, dropout=0.0, metric="queries/sec (lists scored+loss) RankNet sum_session",
                      workload="RankNet pairwise, 256 queries x 64 candidates = 1,032,192 ordered pairs per step "
                               "(BASELINE.json configs[3]; dropout 0 as SURVEY.md 8d)"),
    "evidential600": dict(loss="evidential", cands=64, queries=64, hidden=600, depth=6, task_num=2, last="no_softplus",
                          task_type="evidential_ranking", dropout=0.1,
                          metric="queries/sec (lists scored+loss) UC-Listwise evidential_ranking",
                          workload="UC-Listwise evidential_ranking, D-MPNN depth=6 hidden=600 (BASELINE.json configs[4])"),
}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def under_profiler() -> bool:
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


# ------------------------------------------------------------------------------------------------ shard generation
def _shard_worker(task):
    """Pack steps [lo, hi) into one shard file (runs in a forked CPU-only process, or in-process)."""
    path, lo, hi, seed0, queries, cands, pad_width = task
    from reactranker_amd import featurization, shards, synth
    with shards.ShardWriter(path) as w:
        for i in range(lo, hi):
            qb = synth.make_queries(seed0 + i, queries, cands)
            rb = featurization.BatchMolGraph(qb.r_specs, K=pad_width)
            pb = featurization.BatchMolGraph(qb.p_specs, K=pad_width)
            w.add_step(rb, pb, qb.scope, qb.targets, qb.add_features)
    return path


def build_shards(n_steps, seed0, queries, cands, pad_width, workers):
    """Write `n_steps` distinct packed steps to shard files BEFORE the GPU is touched (worker processes are forked, never
    exec'ed, and stay on the CPU).  Returns (directory, paths, seconds)."""
    t0 = time.time()
    d = tempfile.mkdtemp(prefix="rr_shards_", dir=os.environ.get("RR_SHARD_DIR") or None)
    workers = max(1, min(workers, n_steps))
    per = (n_steps + workers - 1) // workers
    tasks = [(os.path.join(d, f"part{w:02d}.rrshard"), w * per, min(n_steps, (w + 1) * per), seed0, queries, cands, pad_width)
             for w in range(workers) if w * per < n_steps]
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(len(tasks)) as pool:
            paths = pool.map(_shard_worker, tasks)
    else:
        paths = [_shard_worker(t) for t in tasks]
    return d, paths, time.time() - t0


# ------------------------------------------------------------------------------------------------ one configuration
class Runner:
    """Model + optimizer + loss + a pool of packed, HBM-resident steps for one preset."""

    def __init__(self, name, cfg, args, rank, world, local, device, pool_size):
        from reactranker_amd import featurization, synth
        from reactranker_amd import loss as RL
        from reactranker_amd.base_model import build_model
        from reactranker_amd.dp import GradBucket
        from reactranker_amd.train_utils import build_lr_scheduler, build_optimizer
        self.name, self.cfg, self.args = name, cfg, args
        self.rank, self.world, self.local, self.device = rank, world, local, device
        torch.manual_seed(0)                              # identical replicas
        self.model = build_model(hidden_size=cfg["hidden"], mpnn_depth=cfg["depth"], mpnn_diff_depth=cfg["depth"],
                                 ffn_depth=3, use_bias=True, dropout=cfg["dropout"], task_num=cfg["task_num"],
                                 ffn_last_layer=cfg["last"], task_type=cfg["task_type"], add_features_dim=1).to(device)
        self.model.train()
        # the reference's build_optimizer / NoamLR (train/utils.py) through their mirrors; fused Adam = same update, one kernel
        # default: the library's one-launch Adam (HipAdam: torch.optim.Adam's formula and state); --torch-fused-adam /
        # --foreach-adam select torch's own kernels for comparison
        self.opt = build_optimizer(self.model, fused=False if args.foreach_adam else (True if getattr(args, "torch_fused_adam", False) else None))
        self.bucket = GradBucket(self.model.parameters())
        if world > 1:
            self.bucket.attach()                          # gradients are written straight into the all-reduce buffer
        self.sched = build_lr_scheduler(self.opt, warmup_epochs=2, total_epochs=25, train_data_size=100000,
                                        batch_size=cfg["queries"], init_lr=1e-4, max_lr=1e-3, final_lr=1e-4)
        self.RL = RL
        self.mle, self.listnet, self.evid = RL.MLEloss(), RL.ListnetLoss(), RL.evidential_ranking()
        self.pool, self.t_gen, self.t_pack = [], 0.0, 0.0
        for i in range(pool_size):
            t0 = time.time()
            qb = synth.make_queries(1000 * (rank + 1) + i, cfg["queries"], cfg["cands"])
            t1 = time.time()
            rb = featurization.BatchMolGraph(qb.r_specs, K=args.pad_width)      # global pad width (hazard H1)
            pb = featurization.BatchMolGraph(qb.p_specs, K=args.pad_width)
            rb.device_graph(device), pb.device_graph(device)
            ub, _, _ = rb.unique()                            # distinct reactants (used when dropout is inactive)
            ub.device_graph(device)
            t2 = time.time()
            self.t_gen += t1 - t0
            self.t_pack += t2 - t1
            self.pool.append(dict(r=rb, p=pb, scope=qb.scope, targets=torch.tensor(qb.targets).to(device),
                                  add=torch.tensor(qb.add_features).to(device), qb=qb if i == 0 else None))
        self.pairs_per_step = sum(c * (c - 1) for c in self.pool[0]["scope"]) if cfg["loss"] == "ranknet" else None
        gc.collect()
        gc.freeze()                                       # the pool's objects live as long as the run: keep later collections off them

    def loss(self, out, b):
        kind = self.cfg["loss"]
        if kind == "mle":
            return self.mle(out, b["scope"], b["targets"], self.local)
        if kind == "listnet":
            return self.listnet(out, b["scope"], b["targets"], self.local)
        if kind == "evidential":
            return self.evid(out, b["scope"], b["targets"], 1e-4, 0, 1, self.local)
        if kind == "ranknet":                                 # sum_session: loss / ordered pairs of the window
            s, pairs = self.RL.ranknet_loss(out, b["scope"], b["targets"], 1.0, self.local)
            return s / pairs
        raise ValueError(kind)

    def train_step(self, b):
        out = self.model(b["r"], b["p"], gpu=self.local, add_features=b["add"])
        loss = self.loss(out, b)
        self.opt.zero_grad(set_to_none=True)
        self.RL.backward(loss)                            # loss.backward() (train_listwise.py:288) with the constant-one seed
        self.bucket.allreduce(1.0 / self.world)           # equal shards: mean of per-rank normalised grads
        self.sched.step()                                 # NoamLR writes param_groups[0]['lr'] (train/utils.py:88)
        self.opt.step()
        return loss

    def fence(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def settle(self, batches):
        """Untimed steps, once per Runner, in which the library MEASURES which order of weight gradient and input-gradient GEMM this
        workload prefers (functions.WgradOrder: ~14 backward calls, the two orders alternated; a step's results are the same bits in
        either) - a timed region starts in the steady state the rest of a training run sees.  A FIXED number of steps: every
        step ends in a collective, so all ranks must run the same count whatever their own events say."""
        from reactranker_amd import functions as Fn
        if getattr(self, "order_tuning_steps", None) is not None or Fn.WgradOrder.mode != "auto":
            return
        n = Fn.WgradOrder.warm + 2 * Fn.WgradOrder.samples + 6
        for i in range(n):
            self.train_step(batches(i))
            if i % 4 == 3:
                self.fence()                              # (let the events of the last calls complete)
        self.fence()
        Fn.WgradOrder.settled()                           # harvest
        self.order_tuning_steps = n

    def timed(self, batches, n_steps):
        """EXACTLY n_steps optimizer steps between two fences; returns (seconds, per-step device ms list, last loss)."""
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
        self.settle(batches)
        self.fence()
        # The host keeps ~50 k molecule descriptions (Python objects) per step pool: a generation-2 collection that walks them
        # takes tens of milliseconds and lands in whichever step allocates the object that triggers it (seen as ONE 40 ms step in
        # a 200-step streamed leg, about one run in six).  Collector off inside a timed region, as any host-side benchmark would.
        gc_was = gc.isenabled()
        gc.disable()
        try:
            t0 = time.perf_counter()
            marks[0].record()
            last = None
            for i in range(n_steps):
                last = self.train_step(batches(i))
                marks[i + 1].record()
            self.fence()
            secs = time.perf_counter() - t0
        finally:
            if gc_was:
                gc.enable()
        per = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_steps)]
        return secs, per, last


def step_stats(per_ms):
    a = np.asarray(per_ms, np.float64)
    return dict(median=round(float(np.median(a)), 3), min=round(float(a.min()), 3), max=round(float(a.max()), 3),
                note="per-step time between HIP events recorded on the compute stream after every optimizer step")


def csrc_hash():
    h = hashlib.sha256()
    d = os.path.join(REPO, "reactranker_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic():
    """HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/traffic_from_pmc.py); only trusted when
    the kernels' sources have not changed since it was collected (the file records their hash)."""
    try:
        import glob
        tf = sorted(glob.glob(os.path.join(REPO, "profiles", "*_traffic.json")))
        if not tf:
            return {}, None
        d = json.load(open(tf[-1]))
        if d.get("csrc_hash") != csrc_hash():
            return {}, f"{os.path.basename(tf[-1])} is stale (kernel sources changed since it was collected)"
        return {k: (v["hbm_bytes_per_launch"], v.get("launches", 1)) for k, v in d["kernels"].items()}, os.path.basename(tf[-1])
    except Exception as e:                                # noqa: BLE001
        return {}, f"unreadable ({e})"


GATHER_KEYS = ("gather_sum_kernel", "gather_sum_epi_kernel")


def rank_time(key, t):
    """Total time of a kernel key for ranking the GEMM kernels.  A weight-gradient key's event pair spans TWO kernels (the
    GEMM and its fixed-order slab reduce, one C call) and the stream gap between them: ~20 % of its time is not the kernel
    named (rocprofv3 --stats, which sees single kernels, ranks the same way: profiles/r02_bench_kernel_stats.csv)."""
    return t * (0.8 if key.startswith("wgrad") else 1.0)


def summarise(recs, traffic):
    """Per-kernel live timings (HIP events on the launch stream) -> (roofline of the dominant MFMA kernel,
    roofline of the gather kernel, table)."""
    from reactranker_amd import functions as Fn
    roof, roof_g, ktable = None, None, {}
    if not recs:
        return roof, roof_g, ktable

    def tr(key):
        """PMC bytes per launch of a live-timing key.  rocprofv3 names every template instantiation; a key covers the
        instantiations that differ only in trailing parameters the events do not tell apart (the epilogue variant of the
        split GEMM, the addend count of the epilogue gather): launch-weighted mean over them."""
        if key in traffic:
            return traffic[key][0]
        stem = key[:-1] + "," if key.endswith(">") else key + "<"
        hits = [v for k, v in traffic.items() if k.startswith(stem)]
        if key == "gather_sum_kernel":
            hits = [traffic[k] for k in ("gather_sum_kernel<4>",) if k in traffic]
        n = sum(h[1] for h in hits)
        return round(sum(h[0] * h[1] for h in hits) / n) if n else None
    agg = {}
    for rec in recs:                                      # (key, flops, bytes, start event, end event) or (key, flops, bytes, seconds)
        key, flops, nbytes = rec[:3]
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += rec[3] if len(rec) == 4 else rec[3].elapsed_time(rec[4]) * 1e-3
        a[2] += flops
        a[3] += nbytes
    for k, (n, secs, fl, by) in agg.items():
        ktable[k] = dict(launches=n, avg_us=round(secs / n * 1e6, 2), total_ms=round(secs * 1e3, 3),
                         tflops=round(fl / secs / 1e12, 2) if fl else None, algo_gbs=round(by / secs / 1e9, 1))
    mf = {k: v for k, v in agg.items() if v[2] > 0}

    def gemm_roof(dom):
        n, secs, fl, by = mf[dom]
        ach = fl / secs / 1e12
        common = dict(traffic=tr(dom), launches=n, avg_launch_us=round(secs / n * 1e6, 2),
                      algorithmic_flops_per_launch=round(fl / n), algorithmic_bytes_per_launch=round(by / n))
        if "split" in dom:
            # the GEMM issues 6 bf16 MFMA products per f32 multiply: roofline-model intensity = issued flops / algorithmic
            # bytes against the bf16 balance point decides the bound
            prods = F16X2_PRODUCTS if Fn.SplitGemm.f16 else SPLIT_PRODUCTS
            issued = prods * fl
            hbm_bound = issued / max(by, 1.0) < PEAK_BF16_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            mview = dict(f32_equivalent_tflops=round(ach, 2), frac_of_f32_mfma_peak=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                         issued_bf16_tflops=round(issued / secs / 1e12, 1),
                         frac_of_bf16_mfma_peak=round(issued / secs / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                         products_per_multiply=prods,
                         note=("f32 result from two f16 terms per operand (22 significant bits, scaled per tensor), 3 f16 MFMA "
                               "products per multiply (v_mfma_f32_16x16x32_f16)" if Fn.SplitGemm.f16 else
                               "f32 result from three exact bf16 terms per operand, 6 bf16 MFMA products per multiply "
                               "(v_mfma_f32_16x16x32_bf16)") + "; f32-equivalent = algorithmic 2MNK flops")
            gbs = by / secs / 1e9
            hview = dict(algorithmic_gbs=round(gbs, 1), frac_of_hbm_peak=round(gbs / PEAK_HBM_GBS, 4))
            if hbm_bound:
                return dict(kernel=dom, bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=round(gbs / PEAK_HBM_GBS, 4), mfma=mview, **common)
            return dict(kernel=dom, bound="mfma", achieved=round(issued / secs / 1e12, 1), peak=PEAK_BF16_MFMA_TFLOPS,
                        unit="TFLOP/s", frac=round(issued / secs / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4), mfma=mview, hbm=hview,
                        **common)
        return dict(kernel=dom, bound="mfma", achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), **common)
    if mf:
        order = sorted(mf, key=lambda k: -rank_time(k, mf[k][1]))
        roof = gemm_roof(order[0])
        if len(order) > 1:                               # the runner-up among the GEMM kernels that carried events
            roof["second"] = gemm_roof(order[1])
    def gather_roof(key):
        n, secs, fl, by = agg[key]
        ach = by / secs / 1e9
        return dict(kernel=key, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                    frac=round(ach / PEAK_HBM_GBS, 4), traffic=tr(key), launches=n,
                    avg_launch_us=round(secs / n * 1e6, 2), algorithmic_bytes_per_launch=round(by / n))
    # the plain gather-sum (forward aggregates) and the one with the fused mask / residual-sum epilogue (backward): the one
    # with more total time is reported, the other rides along
    gk = sorted((k for k in GATHER_KEYS if k in agg), key=lambda k: -agg[k][1])
    if gk:
        roof_g = gather_roof(gk[0])
        if len(gk) > 1:
            roof_g["other"] = gather_roof(gk[1])
    return roof, roof_g, ktable


def cpu_baseline(args, cfg, qb0):
    """The CPU oracle (vectorised PyTorch-CPU restatement, pinned to the reference by golden vectors) on this box's
    host cores: fwd + ListMLE + bwd + Adam over a bounded sample of the same workload."""
    from oracle import ref_cpu as O
    from reactranker_amd import synth
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = max(1, min(cores, args.cpu_threads))          # a 1-GPU box owns a 16-core share of the host
    torch.set_num_threads(cores)
    shapes = O.model_shapes(cfg["hidden"], cfg["depth"], cfg["depth"], 3, 1, 1, True)
    P = O.params_from_numpy(synth.seeded_weights(shapes, 0), requires_grad=True)
    params = [p for p in P.values() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    mc = dict(depth=cfg["depth"], diff_depth=cfg["depth"], ffn_depth=3, task_type="with_softplus")
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
        out = O.reaction_forward(P, mc, r, p, qb0.add_features[:m], faithful=faithful)
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
               sample=f"{reps} steps x {nq} queries x {cfg['cands']} candidates, fwd+ListMLE+bwd+Adam, "
                      f"vectorised CPU oracle (oracle/ref_cpu.py), torch {torch.__version__} CPU, {cores} threads")
    nqf = min(nq, 16)
    step(True, nqf)                             # warm-up of the faithful variant (first call pays allocator growth)
    tf = step(True, nqf)
    vec["faithful_variant"] = dict(
        value=round(nqf / tf, 3), unit="queries/s", kind=f"port, faithful loops, {nqf}-query step (NOT the {nq}-query step above)",
        sample=f"1 warm step x {nqf} queries: keeps the reference's per-molecule readout loop (models/mpn.py:224-235); its "
               f"backward is quadratic in the step size, so this number falls as the step grows and is not comparable "
               f"with the {nq}-query figures")
    return vec

# ------------------------------------------------------------------------------------------------ the one JSON line
LINE_LIMIT = 6144          # bytes; the driver's capture truncated round 4's 21.5 KB line from the FRONT (metric, value, ... lost)
ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches",
             "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "traffic")
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _pick(d, keys):
    return None if not d else {k: d[k] for k in keys if k in d}


def compact_line(full: dict) -> dict:
    """The driver-facing line: the contract's keys + roofline + cpu_baseline + one number per comparison leg.  `full` is
    everything bench.py measured (it goes to bench_detail.json)."""
    roof = _pick(full.get("roofline"), ROOF_KEYS)
    if roof is not None:
        fr = full["roofline"]
        mv = fr.get("mfma") or {}
        if mv:                                            # the split GEMM's other view, three numbers
            roof["f32_equivalent_tflops"] = mv.get("f32_equivalent_tflops")
            roof["issued_tflops"] = mv.get("issued_bf16_tflops")
            roof["products_per_multiply"] = mv.get("products_per_multiply")
        if fr.get("hbm"):
            roof["algorithmic_gbs"] = fr["hbm"].get("algorithmic_gbs")
        roof["traffic_source"] = fr.get("traffic_source")
    rg = _pick(full.get("roofline_gather"), ROOF_KEYS)
    cpu = _pick(full.get("cpu_baseline"), ("value", "unit", "cores", "kind", "sample"))
    if cpu and isinstance(cpu.get("sample"), str) and len(cpu["sample"]) > 220:
        cpu["sample"] = cpu["sample"][:217] + "..."

    def leg(d, extra=()):
        return _pick(d, ("queries_per_s", "ms_per_step") + tuple(extra))
    ep = full.get("epoch_stream")
    if ep and "skipped" in ep:
        ep_c = {"skipped": str(ep["skipped"])[:120]}
    else:
        ep_c = None if not ep else {"queries_per_s": ep.get("epoch_queries_per_s"), "vs_resident": ep.get("vs_resident"),
                                    "distinct_steps": ep.get("distinct_steps"), "ms_per_step": ep.get("ms_per_step")}
    pre = None
    if full.get("presets"):
        pre = {k: ({"error": str(v["error"])[:100]} if "error" in v else {"value": v.get("value"), "ms_per_step": v.get("ms_per_step")})
               for k, v in full["presets"].items()}
    dp = full.get("dp")
    dp_c = None
    if dp:
        dp_c = {"backend": dp.get("backend"), "bucket_bytes": dp.get("bucket_bytes"), "allreduce_us": dp.get("allreduce_us"),
                "rccl_ranks": [{"rank": r.get("rank"), "device": r.get("device"), "uuid": str(r.get("uuid", ""))[:18]}
                               for r in dp.get("rccl_ranks", [])]}
    val = full.get("validation") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data", "gemm_arithmetic", "config", "step_ms")}
    line.update({"roofline": roof, "roofline_gather": rg, "cpu_baseline": cpu, "gpu_over_cpu": full.get("gpu_over_cpu"),
                 "f32_mfma_path": leg(full.get("f32_mfma_path")),
                 "epoch_stream": ep_c, "presets": pre, "fwd_loss_queries_per_s": full.get("fwd_loss_queries_per_s"),
                 "validation_queries_per_s": val.get("queries_per_s"), "dp": dp_c, "final_loss": full.get("final_loss"),
                 "detail": full.get("detail")})
    for k in ("f16x2_path", "bf16x3_path"):              # the OTHER split arithmetic, with its own dtype string
        if full.get(k):
            line[k] = leg(full[k], ("dtype",))
    return line


def emit(full: dict, write_detail: bool = True) -> str:
    """Write the detail file(s), print the details to stderr and return the compact line as a string <= LINE_LIMIT bytes.
    write_detail = False (--plan-only: a profiling target with nothing but the timed region) leaves an earlier detail file alone."""
    paths = [os.path.join(REPO, "bench_detail.json")] if write_detail else []
    if write_detail and os.path.isdir(os.path.join(REPO, "gpurun_out")):
        paths.append(os.path.join(REPO, "gpurun_out", "bench_detail.json"))
    written = []
    for q in paths:
        try:
            with open(q, "w") as f:
                json.dump(full, f, indent=1)
            written.append(os.path.relpath(q, REPO))
        except OSError as e:                              # read-only checkout: the details still go to stderr
            log(f"could not write {q}: {e}")
    full["detail"] = written[0] if written else None
    line = compact_line(full)
    out = json.dumps(line, separators=(",", ":"))
    for drop in ("validation_queries_per_s", "fwd_loss_queries_per_s", "step_ms", "gemm_arithmetic", "presets", "epoch_stream"):
        if len(out.encode()) <= LINE_LIMIT:
            break
        line.pop(drop, None)                              # (never reached at the sizes tested; the contract's keys go last)
        out = json.dumps(line, separators=(",", ":"))
    for k in ("kernels", "kernels_isolated", "kernels_fwd", "roofline_isolated", "roofline_gather_isolated", "roofline_fwd",
              "roofline_fwd_gather", "validation", "host_prep_s"):
        if full.get(k) is not None:
            log(f"detail {k}: " + json.dumps(full[k]))
    return out


# ------------------------------------------------------------------------------------------------ N > 1 without a launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start `python -m torch.distributed.run` as a CHILD process
    (one rank per GPU; no exec, and this parent has not touched the GPU), relay rank 0's JSON line and return the child's exit
    code.  Matches how the reference is launched on several GPUs (main_ranknet.py:143-160, main.py:145-161: one process per
    device); the driver's own `torch.distributed.run ... bench.py --gpus N` never comes through here (WORLD_SIZE is set)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    log(f"--gpus {n} without a launcher: starting {n} ranks through torch.distributed.run")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                                # rank 0's line; anything else the ranks print to stdout goes to stderr
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log("the ranks exited 0 without printing a result line")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="mle64", choices=sorted(PRESETS), help="headline preset (BASELINE.json configs)")
    ap.add_argument("--queries-per-step", type=int, default=None, help="override the preset's queries per optimizer step")
    ap.add_argument("--cands", type=int, default=None, help="override the preset's candidates per query")
    ap.add_argument("--pool", type=int, default=6, help="distinct pre-packed steps cycled through")
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--dropout", type=float, default=None)
    ap.add_argument("--pad-width", type=int, default=4, help="global a2b pad width K (same on every rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads for the CPU baseline (box share = 16)")
    ap.add_argument("--no-profile", action="store_true", help="skip the live HIP-event kernel timing")
    ap.add_argument("--no-fwd-only", action="store_true", help="skip the forward+loss-only measurement")
    ap.add_argument("--no-presets", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--preset-steps", type=int, default=8)
    ap.add_argument("--no-epoch", action="store_true", help="skip the streamed-from-disk epoch leg")
    ap.add_argument("--epoch-steps", type=int, default=200, help="DISTINCT packed steps written to shard files and streamed")
    ap.add_argument("--shard-workers", type=int, default=8, help="CPU processes packing the shard files")
    ap.add_argument("--no-f32-path", action="store_true", help="skip the exact-f32-MFMA and other-split-arithmetic comparison legs")
    ap.add_argument("--plan-only", action="store_true",
                    help="ONLY the timed region on the step plans (no per-op event passes, comparison legs, presets, epoch leg or CPU "
                         "baseline): what a rocprofv3 --kernel-trace --stats run should profile, so its percentages are a training step's")
    ap.add_argument("--no-side-stream", action="store_true", help="run weight-gradient GEMMs on the main stream")
    ap.add_argument("--no-aux-stream", action="store_true", help="run the reactant encoder on the main stream")
    ap.add_argument("--aux-backward", action="store_true", help="also run the reactant encoder's backward on the aux stream")
    ap.add_argument("--foreach-adam", action="store_true", help="torch's multi-kernel Adam instead of the library's one-launch Adam")
    ap.add_argument("--torch-fused-adam", action="store_true", help="torch's fused Adam instead of the library's one-launch Adam")
    args = ap.parse_args()
    if args.plan_only:
        args.no_profile = args.no_fwd_only = args.no_presets = args.no_epoch = args.no_f32_path = args.no_cpu_baseline = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:   # BEFORE any torch.cuda call: the parent never touches the GPU
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    cfg = dict(PRESETS[args.config])
    for k, v in (("queries", args.queries_per_step), ("cands", args.cands), ("hidden", args.hidden), ("depth", args.depth),
                 ("dropout", args.dropout)):
        if v is not None:
            cfg[k] = v

    # ---- shard files for the epoch leg: packed by forked CPU processes BEFORE this process touches the GPU
    shard_dir, shard_paths, t_shards, epoch_skip = None, None, 0.0, None
    if not args.no_epoch and args.epoch_steps > 0:
        if torch.cuda.device_count() < 1:
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        try:
            # --epoch-steps distinct steps over the whole job: every rank packs and streams its own share
            n_epoch = (args.epoch_steps + world - 1) // world
            workers = 1 if under_profiler() else max(1, min(args.shard_workers, n_epoch))
            log(f"packing {n_epoch} distinct steps per rank into shard files ({workers} CPU process(es))")
            shard_dir, shard_paths, t_shards = build_shards(n_epoch, 50000 + 100000 * rank, cfg["queries"],
                                                            cfg["cands"], args.pad_width, workers)
            log(f"shards ready in {t_shards:.1f}s: {sum(os.path.getsize(p) for p in shard_paths) / 1e9:.2f} GB in {shard_dir}")
        except Exception as e:                            # noqa: BLE001  (e.g. no space left): report, do not hide
            epoch_skip = f"shard generation failed: {type(e).__name__}: {e}"
            log(epoch_skip)
            if shard_dir:
                shutil.rmtree(shard_dir, ignore_errors=True)
            shard_dir = None

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
    from reactranker_amd import dp as DP

    Fn.SideStream.enabled = not args.no_side_stream
    Fn.AuxStream.enabled = not args.no_aux_stream
    if args.aux_backward:
        Fn.AuxStream.backward = True
    traffic, traffic_src = load_traffic()

    log(f"building the step pool of `{args.config}` (synthetic graphs -> native packer -> HBM)")
    R = Runner(args.config, cfg, args, rank, world, local, device, args.pool)
    pool = R.pool
    log(f"pool ready: {len(pool)} steps, gen {R.t_gen:.1f}s pack+upload {R.t_pack:.1f}s; warmup")

    def cyc(off):
        return lambda i: pool[(off + i) % len(pool)]

    if world > 1:
        DP.GradBucket.profile = True                     # HIP events around the collective (reported under "dp")
    for i in range(args.warmup):
        R.train_step(pool[i % len(pool)])
    R.fence()
    DP.GradBucket.events.clear()
    # The timed region issues every step through the step plan (one native call per pass) and records NO kernel events:
    # `value` is the plan path only.  The live roofline timings come from an extra pass right behind it (below).
    log(f"timing {args.steps} steps")
    elapsed, per_ms, last = R.timed(cyc(args.warmup), args.steps)
    records = []
    if not args.no_profile:
        # in-situ pass ON THE SAME PATH: the same steps through the same step plans, same three streams overlapping, with
        # RR_PLAN_TIME - the library records a HIP-event pair on the launch stream around every split-GEMM and gather-sum launch
        # (rr_plan_timing_take).  ~5 us of stream time per timed launch (~45 per step), which is why it is not part of `value`.
        n_prof = max(2, min(6, args.steps // 5))

        def timed_pass(kinds, modes, n):
            Fn.StepPlan.select_timings(kinds, modes)
            Fn.StepPlan.timing = True
            try:
                for i in range(n):
                    R.train_step(pool[(args.warmup + args.steps + i) % len(pool)])
                R.fence()
            finally:
                Fn.StepPlan.timing = False
                Fn.StepPlan.select_timings()
            return Fn.StepPlan.take_timings()
        # every event pair costs its stream ~5 us and loosens the overlap between the streams (a step with all ~45 heavy
        # launches timed reports 150 us for a kernel rocprofv3 sees at 177 us in the untimed step): rank the GEMM keys on two
        # fully timed steps, then time ONE GEMM mode per pass, then the gathers
        ranking = {}
        for key, flops, _, secs in timed_pass(1, 15, 2):
            ranking[key] = ranking.get(key, 0.0) + secs
        order = sorted(ranking, key=lambda k: -ranking[k])
        modes_done = []
        for key in order[:2]:                             # dominant GEMM key and the runner-up: operand mode = third template value
            mode = int(key.split(",")[2])
            if mode not in modes_done:
                modes_done.append(mode)
                records += timed_pass(1, 1 << mode, n_prof)
        records += timed_pass(6, 0, n_prof)               # both gather kinds
        log(f"in-situ kernel timing passes done ({n_prof} plan steps each: GEMM modes {modes_done}, gathers; {len(records)} timed launches)")
    log(f"timed region done: {elapsed / max(1, args.steps) * 1e3:.2f} ms/step")
    loss_val = float(last.detach().sum().cpu()) if last is not None else float("nan")
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    dp_info = None
    if world > 1:
        ids = [None] * world
        props = torch.cuda.get_device_properties(device)
        dist.all_gather_object(ids, dict(rank=rank, local_rank=local, device=torch.cuda.current_device(),
                                         name=props.name, uuid=str(getattr(props, "uuid", "")), host=os.uname().nodename))
        evs = DP.GradBucket.events
        us = [a.elapsed_time(b) * 1e3 for a, b in evs]
        dp_info = dict(backend=dist.get_backend(), rccl_ranks=ids, bucket_bytes=R.bucket.numel * 4,
                       allreduce_us=dict(median=round(float(np.median(us)), 1), min=round(min(us), 1), max=round(max(us), 1),
                                         calls=len(us)) if us else None,
                       note="one all-reduce of the flat fp32 gradient bucket per step, HIP events on the compute stream "
                            "around the collective (rank 0)")
        DP.GradBucket.profile = False

    extra = {}
    if not args.no_fwd_only:                              # report fwd+loss and fwd+loss+bwd separately
        from reactranker_amd import eval as RE
        R.model.eval()
        with torch.no_grad():                             # upload the de-duplication maps outside the timed loop
            for b in pool:
                R.model(b["r"], b["p"], gpu=local, add_features=b["add"])
        R.fence()
        tf0 = time.perf_counter()
        with torch.no_grad():
            for i in range(args.steps):
                b = pool[i % len(pool)]
                R.loss(R.model(b["r"], b["p"], gpu=local, add_features=b["add"]), b)
        R.fence()
        extra["fwd_loss_queries_per_s"] = round(world * args.steps * cfg["queries"] / (time.perf_counter() - tf0), 1)
        extra["fwd_loss_note"] = "eval mode (no dropout): forward + loss only; reactant encoder runs once per distinct reactant"
        # the per-epoch validation of the reference (train/eval.py:475-555: one forward per query + Python lists) as it runs
        # here: one forward per 64-query step + ONE rr_ranking_metrics_f32 launch (all 12 statistics of every query)
        with torch.no_grad():
            for i in range(2):
                b = pool[i % len(pool)]
                RE.ranking_stats(R.model(b["r"], b["p"], gpu=local, add_features=b["add"]), b["scope"], b["targets"], local)
        R.fence()
        mev = []
        tv0 = time.perf_counter()
        with torch.no_grad():
            for i in range(args.steps):
                b = pool[i % len(pool)]
                out = R.model(b["r"], b["p"], gpu=local, add_features=b["add"])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                RE.ranking_stats(out, b["scope"], b["targets"], local)
                e1.record()
                mev.append((e0, e1))
        R.fence()
        tv = time.perf_counter() - tv0
        mus = [a.elapsed_time(b_) * 1e3 for a, b_ in mev]
        extra["validation"] = dict(
            queries_per_s=round(world * args.steps * cfg["queries"] / tv, 1), ms_per_step=round(tv / args.steps * 1e3, 3),
            ranking_metrics_launch_us=dict(median=round(float(np.median(mus)), 1), min=round(min(mus), 1), max=round(max(mus), 1)),
            note="eval-mode forward of a 64-query step + one rr_ranking_metrics_f32 launch (stable sort by score and by target, "
                 "top-1 / top-25% / recall / NDCG / evaluate_top_scores / calculate_ndcg statistics of every query, one "
                 "wavefront per query); HIP events around the metrics launch on the compute stream")
        if not args.no_profile:
            # roofline of the forward-only path: per-op issue with HIP events around every heavy launch (eval mode, reactant
            # de-duplication on, the product / distinct-reactant encoders overlapping on two streams as in production)
            Fn.Profiler.start()
            with torch.no_grad():
                for i in range(min(args.steps, 6)):
                    b = pool[i % len(pool)]
                    R.model(b["r"], b["p"], gpu=local, add_features=b["add"])
            R.fence()
            rf, rfg, ktf = summarise(Fn.Profiler.stop(), {})
            if rf:
                rf["note"] = ("eval-mode forward only (what validation and inference run): HIP events around every heavy launch, "
                              "issued per op; traffic not collected for this pass")
            extra["roofline_fwd"], extra["roofline_fwd_gather"], extra["kernels_fwd"] = rf, rfg, ktf
        R.model.train()

    # timed region: kernels of the three streams overlap, so a kernel's launch duration includes the time it
    # shares the chip with kernels of the other streams (this is what rocprofv3 --stats of this command shows)
    roof, roof_g, ktable = summarise(records, traffic)
    load_clock = {"clock_ghz": 1.8, "peak": round(PEAK_F32_MFMA_TFLOPS * 1.8 / 2.4, 1), "unit": "TFLOP/s",
                  "source": "shader clock measured inside the k-loop of this kernel under load (s_memtime / s_memrealtime, "
                            "tools/trace_linear.py, profiles/r01_linear_phase_trace.txt); the datasheet peak assumes 2.4 GHz"}
    if roof:
        if roof["bound"] == "mfma" and "split" not in roof["kernel"]:
            roof["peak_at_load_clock"] = load_clock
        roof["traffic_source"] = traffic_src
        roof["note"] = ("extra pass right behind the timed region on the SAME path (step plans with RR_PLAN_TIME: same steps, same "
                        "model state, same three streams overlapping), HIP events on the launch stream around every launch of "
                        "this kernel (rr_plan_timing_take); the timed region itself carries no events")
    # isolated pass: the same steps with every kernel serialised on one stream -> per-kernel quality
    roof_iso = roof_g_iso = None
    ktable_iso = {}
    if records:
        side0, aux0 = Fn.SideStream.enabled, Fn.AuxStream.enabled
        Fn.SideStream.enabled = Fn.AuxStream.enabled = False
        R.fence()
        Fn.Profiler.start()
        for i in range(min(args.steps, 10)):
            R.train_step(pool[(args.warmup + args.steps + i) % len(pool)])
        R.fence()
        roof_iso, roof_g_iso, ktable_iso = summarise(Fn.Profiler.stop(), traffic)
        Fn.SideStream.enabled, Fn.AuxStream.enabled = side0, aux0
        if roof_iso:
            if roof_iso["bound"] == "mfma" and "split" not in roof_iso["kernel"]:
                roof_iso["peak_at_load_clock"] = load_clock
            roof_iso["note"] = "extra pass after the timed region, one stream (kernels do not overlap)"

    # ---- the same step with every GEMM on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the split path's reference point
    f32_path = None
    if not args.no_f32_path:
        Fn.SplitGemm.enabled = False
        try:
            for i in range(3):
                R.train_step(pool[i % len(pool)])
            n32 = min(args.steps, 15)
            s32, per32, _ = R.timed(cyc(args.warmup + args.steps), n32)
            t32 = torch.tensor([s32], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(t32, op=dist.ReduceOp.MAX)
            f32_path = dict(queries_per_s=round(world * n32 * cfg["queries"] / float(t32.item()), 2),
                            ms_per_step=round(float(t32.item()) / n32 * 1e3, 3), steps=n32, step_ms=step_stats(per32),
                            note="same model state, same steps, SplitGemm.enabled = False (RR_PLAN_F32_GEMM): every GEMM on "
                                 "v_mfma_f32_16x16x4_f32; `value` above is the " +
                                 ("two-f16-term path" if Fn.SplitGemm.f16 else "three-exact-bf16-term path"))
        finally:
            Fn.SplitGemm.enabled = True
        log(f"f32-MFMA path: {f32_path}")
    # ---- ... and on the OTHER split arithmetic: the opt-in two-f16-term form beside the default three-exact-bf16-term headline
    # (or the other way round when RR_F16X2=1 made two terms this run's arithmetic)
    other_path, other_key = None, ("bf16x3_path" if Fn.SplitGemm.f16 else "f16x2_path")
    if not args.no_f32_path:
        f16_was = Fn.SplitGemm.f16
        Fn.SplitGemm.f16 = not f16_was
        try:
            for i in range(3):
                R.train_step(pool[i % len(pool)])
            n3 = min(args.steps, 15)
            s3, per3, _ = R.timed(cyc(args.warmup + args.steps), n3)
            t3 = torch.tensor([s3], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            other_path = dict(queries_per_s=round(world * n3 * cfg["queries"] / float(t3.item()), 2),
                              ms_per_step=round(float(t3.item()) / n3 * 1e3, 3), steps=n3, step_ms=step_stats(per3),
                              dtype=DTYPE_BF16X3 if f16_was else DTYPE_F16X2,
                              note="same model state, same steps, SplitGemm.f16 = " + str(not f16_was) + ": encoder GEMMs and weight "
                                   "gradients on " + ("three exact bf16 terms per operand (6 x v_mfma_f32_16x16x32_bf16 per k-step)"
                                                      if f16_was else
                                                      "two f16 terms per operand scaled per tensor (3 x v_mfma_f32_16x16x32_f16 per k-step; "
                                                      "22-bit operands, batch-dependent below 2^-40 of a tensor's largest magnitude: opt-in)"))
        finally:
            Fn.SplitGemm.f16 = f16_was
        log(f"{other_key}: {other_path}")

    # ---- streamed epoch: the SAME training step fed from shard files on disk
    epoch = None
    if shard_dir is not None:
        from reactranker_amd import shards as SH
        try:
            rd = SH.ShardSet(shard_paths)
            n = len(rd)
            log(f"streaming {n} distinct steps from {len(shard_paths)} shard file(s)")
            # Two untimed steps through the streaming pipeline first, like the warm-up of the resident region: the first
            # streamed step pays for what a resident step never needs - the device-side f_bonds rebuild and the prefetcher's
            # own staging buffers come out of the caching allocator (a fresh hipMalloc when its cached blocks have just been
            # handed to the prefetcher), the reader thread makes its first HIP calls - 37 / 87 / ~95 ms on three boxes, which
            # was ALL of the 0.925-0.975x the 200-step leg showed (its per-step median was already below the resident
            # region's).  A real epoch is 1563 steps: its first step is noise there, it was 8 % of a 200-step leg.
            # ... and they are the epoch's LARGEST steps (ShardSet.largest): the caching allocator then holds a block for every
            # later, smaller request - without this 4 device allocations (126 MB) fell inside the 200 timed steps, each a
            # hipMalloc that usually costs a millisecond and once in a while 50 (one run of six: a 58 ms step, 0.937x)
            warm = rd.largest(min(2, n))
            n_warm = len(warm)
            pf = SH.StepPrefetcher(rd, device, warm + list(range(n)), depth=3)
            startup = pf.prime()                          # start-up latency of the pipeline, reported, not timed
            it = iter(pf)
            for _ in range(n_warm):                       # the SAME pipeline: its buffers, its thread, its copy stream are
                R.train_step(next(it))                    # in steady state when the timed epoch starts
            R.fence()
            ms0 = torch.cuda.memory_stats(device)
            ws0 = Fn._WorkspacePool.allocs
            e_secs, e_per, _ = R.timed(lambda i: next(it), n)
            ms1 = torch.cuda.memory_stats(device)
            # device allocations INSIDE the timed epoch (a fresh hipMalloc is 40-180 ms on this pool: the slowest step, if any)
            mem = dict(device_allocs=int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                       reserved_mb_grown=round((ms1.get("reserved_bytes.all.current", 0) - ms0.get("reserved_bytes.all.current", 0)) / 1e6, 1),
                       workspace_allocs=int(Fn._WorkspacePool.allocs - ws0))
            pf.close()
            te = torch.tensor([e_secs], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
            e_secs = float(te.item())
            qps_e = world * n * cfg["queries"] / e_secs
            epoch = dict(epoch_queries_per_s=round(qps_e, 2), ms_per_step=round(e_secs / n * 1e3, 3), distinct_steps=n * world,
                         distinct_steps_per_rank=n,
                         vs_resident=round(qps_e / (world * args.steps * cfg["queries"] / elapsed), 4),
                         step_ms=step_stats(e_per), slowest_steps=[[int(i), round(float(e_per[i]), 2)] for i in np.argsort(e_per)[-3:][::-1]],
                         shard_files=len(shard_paths),
                         shard_gb_per_rank=round(sum(os.path.getsize(p) for p in shard_paths) / 1e9, 3),
                         step_mb=round(rd.max_step_bytes / 1e6, 2), h2d_gb_per_rank=round(pf.bytes_copied / 1e9, 3),
                         consumer_wait_s=round(pf.wait_s, 4), prefetch_startup_s=round(startup, 4), pack_s=round(t_shards, 2),
                         page_cache="hot", warmup_steps=min(2, n), allocator=mem,
                         note="every step read once from shard files written seconds earlier by this run, i.e. served from the "
                              "page cache, not from the disk (page cache -> pinned staging -> one H2D copy per step on a "
                              "copy stream, 3 slots); only the 22 bond columns of f_bonds and the distinct reactants' "
                              "features travel, the rest is rebuilt on the device; same model / optimizer state continues "
                              "from the timed region")
            log(f"epoch stream: {e_secs / n * 1e3:.2f} ms/step, {qps_e:.0f} queries/s")
            log("epoch stream detail: " + json.dumps({k: epoch[k] for k in ("vs_resident", "step_ms", "slowest_steps", "consumer_wait_s",
                                                                           "prefetch_startup_s", "h2d_gb_per_rank", "allocator")}))
        except Exception as e:                            # noqa: BLE001
            epoch = dict(skipped=f"{type(e).__name__}: {e}")
            log(f"epoch stream failed: {epoch['skipped']}")
        finally:
            shutil.rmtree(shard_dir, ignore_errors=True)
    elif epoch_skip:
        epoch = dict(skipped=epoch_skip)

    # ---- the other BASELINE configurations at full step size
    presets = {}
    if not args.no_presets:
        for name in PRESETS:
            if name == args.config:
                continue
            pc = PRESETS[name]
            try:
                log(f"preset {name}: building 2 steps")
                del_R = Runner(name, pc, args, rank, world, local, device, 2)
                for i in range(6):                        # un-synchronised, like the timed loop: lets the caching
                    del_R.train_step(del_R.pool[i % 2])   # allocator reach its steady state (no hipMalloc while timing)
                secs, per, lastp = del_R.timed(lambda i: del_R.pool[i % 2], args.preset_steps)
                tp = torch.tensor([secs], dtype=torch.float64, device=device)
                if world > 1:
                    dist.all_reduce(tp, op=dist.ReduceOp.MAX)
                secs = float(tp.item())
                roofp = None
                if not args.no_profile:                   # separate short pass with events on every heavy launch
                    Fn.Profiler.start()
                    for i in range(3):
                        del_R.train_step(del_R.pool[i % 2])
                    del_R.fence()
                    roofp, roofg, _ = summarise(Fn.Profiler.stop(), {})
                    if roofp:
                        roofp["note"] = "3 extra steps with HIP events around every heavy launch; streams overlap"
                g0 = del_R.pool[0]["p"].device_graph(device)
                presets[name] = dict(
                    metric=pc["metric"], value=round(world * args.preset_steps * pc["queries"] / secs, 2), unit="queries/s",
                    ms_per_step=round(secs / args.preset_steps * 1e3, 3), steps=args.preset_steps, step_ms=step_stats(per),
                    config=dict(workload=pc["workload"], queries_per_step_per_gpu=pc["queries"],
                                candidates_per_query=pc["cands"], hidden=pc["hidden"], depth=pc["depth"],
                                task_num=pc["task_num"], dropout=pc["dropout"], atoms_per_step_side=int(g0.nA),
                                directed_bonds_per_step_side=int(g0.nB),
                                ordered_pairs_per_step=del_R.pairs_per_step),
                    roofline=roofp, roofline_gather=roofg if not args.no_profile else None,
                    final_loss=round(float(lastp.detach().sum().cpu()), 6))
                log(f"preset {name}: {presets[name]['ms_per_step']} ms/step")
                del del_R
            except Exception as e:                        # noqa: BLE001
                presets[name] = dict(error=f"{type(e).__name__}: {e}")
                log(f"preset {name} failed: {presets[name]['error']}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and cfg["loss"] == "mle":
        log("CPU baseline (oracle) on the host cores")
        cpu = cpu_baseline(args, cfg, pool[0]["qb"])
        log("CPU baseline done")

    if rank == 0:
        qps = world * args.steps * cfg["queries"] / elapsed
        g0 = pool[0]["p"].device_graph(device)
        full = {
            "metric": cfg["metric"], "value": round(qps, 2), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_F16X2 if Fn.SplitGemm.f16 else DTYPE_BF16X3, "data": "synthetic",
            "gemm_arithmetic": ("f32 in/out/accumulate; encoder GEMMs via two f16 terms per operand scaled per tensor (22-bit operands, "
                                "3 MFMA products); FFN head on the f32 MFMA" if Fn.SplitGemm.f16 else
                                "f32 in/out/accumulate; encoder GEMMs via three EXACT bf16 terms per operand (no operand bit dropped, "
                                "6 MFMA products); FFN head on the f32 MFMA"),
            "config": {"workload": f"{cfg['workload']}, {cfg['queries']}-query steps, D-MPNN depth={cfg['depth']} "
                                   f"hidden={cfg['hidden']}",
                       "preset": args.config,
                       "phase": "train step: fwd + loss + bwd + grad all-reduce + Adam/NoamLR",
                       "queries_per_step_per_gpu": cfg["queries"], "candidates_per_query": cfg["cands"],
                       "atoms_per_step_side": int(g0.nA), "directed_bonds_per_step_side": int(g0.nB),
                       "pad_width_K": int(g0.K), "dropout": cfg["dropout"], "step_pool": len(pool),
                       "parallelism": f"dp{world}"},
            "step_ms": {k: v for k, v in step_stats(per_ms).items() if k != "note"},
            "roofline": roof, "roofline_gather": roof_g, "roofline_isolated": roof_iso,
            "roofline_gather_isolated": roof_g_iso, "cpu_baseline": cpu,
            "f32_mfma_path": f32_path, other_key: other_path, "epoch_stream": epoch, "presets": presets or None, "dp": dp_info,
            "kernels": ktable, "kernels_isolated": ktable_iso, "final_loss": round(loss_val, 6),
            "host_prep_s": {"synthetic_generation": round(R.t_gen, 2), "native_pack_and_upload": round(R.t_pack, 2)},
            "wgrad_order": {"mode": Fn.WgradOrder.mode, "untimed_tuning_steps": getattr(R, "order_tuning_steps", None) or 0,
                            "early_by_workload": Fn.WgradOrder.choices(),
                            "note": "order of a layer's weight gradient and input-gradient GEMM, measured per workload in untimed steps "
                                    "before each timed region (functions.WgradOrder); results are the same bits in either order"},
        }
        full.update(extra)
        if cpu:
            full["gpu_over_cpu"] = round(qps / cpu["value"], 1)
        print(emit(full, write_detail=not args.plan_only), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
