"""Headline benchmark: GPSA training steps/sec on the 2-view x 10k-spot, M=200, 50-output grid.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          (starts its own N ranks through torch.distributed.run), or under a launcher:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = forward(S) + loss_fn + backward + Adam update (the reference loop,
examples/grid_example.py:62-78) on synthetic inputs already resident in HBM.  N > 1: the rows of
every view are sharded over the ranks (strong scaling: total problem fixed), one RCCL all-reduce of the
flattened gradient per step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md:42 (fp32-input MFMA = fp32 vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--S", type=int, default=5, help="MC samples per step (reference training value)")
    ap.add_argument("--side", type=int, default=100, help="grid side per view (100 -> 10k spots)")
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--outputs", type=int, default=50)
    ap.add_argument("--M", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="disable the per-forward numerics sync")
    ap.add_argument("--no-graph", action="store_true", help="skip the extra hipGraph-replay timing")
    ap.add_argument("--no-s1", action="store_true", help="skip the secondary S = 1 timing")
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed blocks of --steps steps each (barrier + synchronize around every block); value = steps / "
                         "median block, min / max reported")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the config1 / config3 extra keys (BASELINE configs 1 and 3 on this GPU)")
    ap.add_argument("--headline-only", action="store_true",
                    help="profiling: only the headline blocks (no helper / verbatim / exact-mode / S = 1 / graph / config "
                         "legs, no CPU baseline)")
    ap.add_argument("--overlap-reduce", action="store_true",
                    help="N > 1: reduce the data GP's span of the gradient buffer on a side stream while the rest of the "
                         "backward runs (parallel.GradAllReducer(overlap=True)); off by default: never run on N > 1 "
                         "distinct devices by the builder")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the live rocprofv3 --pmc passes for roofline.traffic (the committed file is quoted instead)")
    ap.add_argument("--graph-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--static-grads", action="store_true", help="diagnostic: zero_grad(set_to_none=False)")
    ap.add_argument("--overlap", action="store_true", help="diagnostic: per-view side streams in eager mode too")
    ap.add_argument("--workload", choices=["1", "2", "3", "4", "5"], default="2",
                    help="BASELINE.json configuration (1-based).  2 = the contract metric's; 1 = the reference example's "
                         "size (2 x 100 spots, 30 outputs, M = 25, view 0 fixed), 3 = 4 views x 10k spots, 500 outputs "
                         "through 10 latent GPs, Matern-1/2 warp; 4 / 5 = the Visium- / Slide-seq-scale configurations at "
                         "their stated size (S = 1, independent outputs).  All but 2: diagnostic lines")
    ap.add_argument("--shard", choices=["rows", "outputs"], default="rows",
                    help="N > 1: rows of every view (default, strong scaling of the contract metric) or the output "
                         "axis (parallel.shard_outputs: per-output parameters and gradients never leave their rank)")
    ap.add_argument("--kl-share", choices=["owner", "scaled"], default="owner",
                    help="row-sharded ranks: 'owner' = each rank evaluates its own range of the KL terms at weight 1 "
                         "(factorises only its variational covariances; the all-reduce sums the shares), 'scaled' = "
                         "every rank evaluates all of them at weight 1/world (rounds 1-5)")
    ap.add_argument("--sustained", type=int, default=1000,
                    help="extra (default line only): that many consecutive headline steps after the timed blocks, with "
                         "the clock and power rocm-smi saw meanwhile (0: skip)")
    ap.add_argument("--emulate-shard", type=int, default=1,
                    help="diagnostic: time rank 0's share of a K-way row sharding on ONE GPU (no all-reduce); "
                         "the line is then NOT the contract metric")
    args = ap.parse_args()
    args.fixed, args.latent, args.warp = None, None, "rbf"
    if args.headline_only:
        args.no_s1 = args.no_graph = args.no_extras = args.no_cpu_baseline = True
    if args.workload == "1":    # examples/grid_example.py: 2 views x 100 spots, 30 outputs, M = 25, fixed_view_idx = 0
        args.side, args.views, args.outputs, args.M, args.fixed = 10, 2, 30, 25, 0
    elif args.workload == "3":  # 4 views x 10k spots, 500 outputs through 10 latent GPs, Matern-1/2 warp, M = 200
        args.side, args.views, args.outputs, args.M, args.latent, args.warp = 100, 4, 500, 200, 10, "matern12"
    if args.workload == "4":    # 8 views x 5041 spots, 2000 genes, M = 500, fixed_view_idx = 0
        args.side, args.views, args.outputs, args.M, args.S, args.fixed = 71, 8, 2000, 500, 1, 0
    elif args.workload == "5":  # 2 views x 99 856 spots, 1000 genes, M = 1000
        args.side, args.views, args.outputs, args.M, args.S = 316, 2, 1000, 1000, 1
    if args.workload != "2":
        args.no_s1 = args.no_extras = True
        args.no_cpu_baseline = args.workload != "1"  # (config 1 is BASELINE's "CPU reference" configuration)
        args.no_graph = args.workload != "1"
    return args


def step_graph_stats(model):
    """[replays, eager calls, captures, graphs held] of the engine's hipGraph cache over the model's plans (on by default
    for launch-bound plans only: GPSA_STEP_GRAPH=0 / 1 overrides)"""
    import ctypes

    tot = [0, 0, 0, 0]
    try:
        for plan in model.__dict__.get("_step_plans", {}).values():
            st = (ctypes.c_longlong * 4)()
            if plan.lib.gpsa_step_graph(plan.handle, -1, st) == 0:
                tot = [a + int(b) for a, b in zip(tot, st)]
    except Exception as e:  # (diagnostic only)
        return dict(error=f"{type(e).__name__}: {e}"[:200])
    return dict(replays=tot[0], eager_calls=tot[1], captures=tot[2], graphs_held=tot[3],
                env=os.environ.get("GPSA_STEP_GRAPH", "unset (off)"))


_GC_DONE = []  # (time_blocks: one full collection + freeze before the first timed block of the process)


class KernelTimer:
    """HIP-event timing of the three contraction launches, recorded by the step engine itself on the stream it
    launches them on (gpsa_step_timing: events around gpsa_quadform_fwd / _bwd_alpha / _bwd_omega of the data
    GP), for every step of the timed region."""

    NAMES = ("quadform_fwd", "quadform_bwd_alpha", "quadform_bwd_omega")

    def __init__(self, model, steps):
        self.model, self.steps, self.plans = model, steps, []

    def start(self):
        for plan in self.model.__dict__.get("_step_plans", {}).values():
            if plan.S == self.S and plan.lib.gpsa_step_timing(plan.handle, self.steps) == 0:
                self.plans.append(plan)

    def summary(self, M, C, L):
        import ctypes

        out = {}
        for plan in self.plans:
            buf = (ctypes.c_float * (3 * self.steps))()
            n = plan.lib.gpsa_step_timing_read(plan.handle, buf, self.steps)
            plan.lib.gpsa_step_timing(plan.handle, 0)
            if n <= 0:
                continue
            flops = 2.0 * M * M * C * L
            for k, name in enumerate(self.NAMES):
                ms = [buf[i * 3 + k] for i in range(n)]
                out[name] = dict(launches=n, avg_ms=sum(ms) / n, flops=flops,
                                 tflops=flops * n / (sum(ms) * 1e-3) / 1e12)
        return out


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, state, dd_cpu):
    """The oracle (op-for-op PyTorch-CPU restatement of the reference) timed on this host: 1 warm-up step,
    then the median of 3 timed steps, each = forward + ELBO + backward + a torch.optim.Adam update of the
    same parameters (the reference loop, examples/grid_example.py:62-78)."""
    import psutil

    from oracle import gpsa_oracle as orc

    torch.autograd.set_detect_anomaly(False)  # the shipped reference turns it ON (1.3-1.4x slower)
    m = "expression"
    N, L = dd_cpu[m]["spatial_coords"].shape[0], args.outputs
    avail = psutil.virtual_memory().available / 2**30
    need = 6.0 * args.S * L * N * args.M * 4 / 2**30  # ~22 GB at the headline config
    S_run = args.S if avail > need * 1.3 else 1
    cfg = dict(modality_names=[m], n_views=args.views, n_spatial_dims=2, kernel_warp="rbf",
               kernel_data="rbf", n_latent_gps={m: None}, fixed_view_idx=args.fixed)
    gen = torch.Generator().manual_seed(1)
    n_v = N // args.views
    ns = {m: dd_cpu[m]["n_samples_list"]}
    X, Y = {m: dd_cpu[m]["spatial_coords"]}, {m: dd_cpu[m]["outputs"]}
    params = {k: torch.nn.Parameter(v.clone()) for k, v in state.items() if k.startswith(orc.TRAINABLE_PREFIXES)}
    opt = torch.optim.Adam(list(params.values()), lr=1e-2)
    times, loss = [], None
    n_free = args.views - (0 if args.fixed is None else 1)
    n_timed = 3 if N * L > 1e5 else 20  # (the reference example's size: milliseconds per step)
    for it in range(1 + n_timed):  # 1 warm-up + the timed ones
        eps_G = [torch.randn(S_run, n_v, 2, generator=gen) for _ in range(n_free)]
        eps_F = {m: torch.randn(S_run, N, L, generator=gen)}
        st = dict(state)
        st.update({k: p.detach() for k, p in params.items()})
        t0 = time.time()
        r = orc.evaluate(st, cfg, X, Y, ns, S_run, eps_G, eps_F, dtype=torch.float32)
        for k, p in params.items():
            p.grad = r["grads"][k]
        opt.step()
        dt = time.time() - t0
        loss = r["loss"]
        assert torch.isfinite(loss)
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    scale = args.S / S_run
    return dict(
        value=1.0 / (med * scale), unit="steps/s (forward+ELBO+backward+Adam)",
        cores=torch.get_num_threads(), nproc=os.cpu_count(), cpu_model=_cpu_model(), kind="port",
        sample=f"oracle steps at the full config with S={S_run}: 1 warm-up, then median of {n_timed} timed "
               + ("[3 of the 20 steps BASELINE.md section 3 planned: a step is ~11 s here, 20 would not fit the default "
                  "run's few minutes] " if n_timed == 3 else "")
               + f"({', '.join(f'{t:.3g}' for t in times[:5])}{' ...' if len(times) > 5 else ''} s)"
               + (f", time scaled x{scale:.0f} to S={args.S}" if scale != 1 else "")
               + f"; anomaly mode off, host RAM avail {avail:.0f} GB",
    )


def parity_at_bench_size(args, model, dd_cpu, dd, view_idx, Ns):
    """ONE TRAINING STEP of the TRAINED bench model at the full bench size AND THE TIMED S (5: the timed launch geometry,
    C = S N = 100 000 columns) - the reference's two calls, i.e. the fused ELBO step the timed loop runs
    (panel_elbo_kernel: column tiles split over workgroups, partial tiles through the slabs), then backward, in the
    model's timed configuration (exact_inducing_grad as timed) - against the fp64 oracle on the same parameters and
    the same injected draws: outputs, ELBO and every parameter gradient (the fp32 oracle the timing leg runs is no
    yardstick here: at M = 200 the fp32 reference is 1e-1 from its own fp64 run on F).  Norm-wise relative
    errors.  The fp64 oracle holds the reference's [S, L, N, M] tensor several times over (8 GB a copy at S = 5): with
    less than ~80 GB of host memory free the leg falls back to S = 1 and says so."""
    import psutil

    from oracle import gpsa_oracle as orc

    m = "expression"
    t0 = time.time()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for name in ("mean_slopes", "mean_intercepts"):
        state.setdefault(name, getattr(model, name).detach().cpu().clone())
    N, L = dd_cpu[m]["spatial_coords"].shape[0], args.outputs
    need_gb = 10.0 * args.S * L * N * args.M * 8 / 2**30
    S = args.S if psutil.virtual_memory().available / 2**30 > need_gb else 1
    gen = torch.Generator().manual_seed(7)
    eps_G = [torch.randn(S, n_v, 2, generator=gen) for n_v in dd_cpu[m]["n_samples_list"]]
    eps_F = {m: torch.randn(S, N, L, generator=gen)}
    dev = model.Xtilde.device
    model.inject_noise([e.to(dev) for e in eps_G], {m: eps_F[m].to(dev)}, None)
    model.zero_grad(set_to_none=True)
    out = model.forward({m: dd[m]["spatial_coords"]}, view_idx=view_idx, Ns=Ns, S=S)
    loss = model.loss_fn(dd, out[3])
    loss.backward()
    fuse = getattr(model._cache, "fuse", None)
    exact = bool(model._cache.plan.exact)  # (the plan this very forward ran)
    cfg = dict(modality_names=[m], n_views=args.views, n_spatial_dims=2, kernel_warp="rbf", kernel_data="rbf",
               n_latent_gps={m: None}, fixed_view_idx=None)
    ref = orc.evaluate(state, cfg, {m: dd_cpu[m]["spatial_coords"]}, {m: dd_cpu[m]["outputs"]},
                       {m: dd_cpu[m]["n_samples_list"]}, S, eps_G, eps_F, want_grads=True, dtype=torch.float64)
    rel = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / b.double().norm())
    gerr = {k: rel(p.grad, ref["grads"][k]) for k, p in model.named_parameters()
            if p.grad is not None and k in ref["grads"] and float(ref["grads"][k].norm()) > 0}
    res = dict(S=S, S_timed=args.S, exact_inducing_grad=exact,
               step="forward + loss_fn + backward, " + ("fused ELBO (panel_elbo_kernel)" if fuse is not None and
               "fused" in fuse["state"] else "separate kernels"),
               G_means_rel=rel(out[0][m], ref["G_means"][m]), G_samples_rel=rel(out[1][m], ref["G_samples"][m]),
               F_rel=rel(out[3][m], ref["F_obs"][m]), loss_rel=rel(loss.reshape(1), ref["loss"].reshape(1)),
               grad_rel_max=max(gerr.values()), grad_rel_worst=max(gerr, key=gerr.get),
               grad_rel={k: float(f"{v:.2e}") for k, v in gerr.items()},
               # the scalar hyper-parameters' gradients themselves, [this step, fp64 oracle]: a relative error on a
               # scalar that is passing through zero (a trained model's stationary point) says little by itself
               grad_scalars={k: [p.grad.detach().cpu().double().reshape(-1).tolist(), ref["grads"][k].reshape(-1).tolist()]
                             for k, p in model.named_parameters()
                             if p.grad is not None and k in ref["grads"] and p.numel() <= 2},
               tolerance=1e-4, against="oracle/gpsa_oracle.py in fp64, same trained parameters, same injected draws",
               seconds=round(time.time() - t0, 1))
    model.zero_grad(set_to_none=True)
    return res


def measure_traffic(args):
    """HBM-side bytes per launch of the contraction kernels, MEASURED NOW: two child runs of this file's headline loop
    under ``rocprofv3 --pmc`` (FETCH_SIZE, then WRITE_SIZE: separate passes, as MI355X_MICROARCH.md's HBM section
    prescribes), summarised by profiles/pmc_summarize.py's rules (KiB -> bytes, FETCH doubled on gfx950).  None when
    rocprofv3 is not on this box or a pass fails (the line then falls back to the committed profiles/pmc_traffic.json
    and says so)."""
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None  # this process is itself being profiled: no profiler inside the profiler
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        import pmc_summarize as PS

        t0 = time.time()
        tmp = tempfile.mkdtemp(prefix="gpsa_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        acc = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--headline-only", "--blocks", "1", "--steps", "3", "--warmup", "1",
                   "--S", str(args.S), "--side", str(args.side), "--views", str(args.views), "--outputs",
                   str(args.outputs), "--M", str(args.M)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return dict(error=f"rocprofv3 --pmc {counter}: rc={r.returncode}", stderr=r.stderr[-300:])
            acc[counter] = PS.per_kernel(d, counter)
        out = PS.summarise(acc["FETCH_SIZE"], acc["WRITE_SIZE"])
        out["seconds"] = round(time.time() - t0, 1)
        shutil.rmtree(tmp, ignore_errors=True)
        return out
    except Exception as e:  # the contract line must not die with its extra
        return dict(error=f"{type(e).__name__}: {e}"[:300])


def split_bf16_experiment():
    """EXPLORATORY extra (never the headline; the timed dtype stays f32): the three-way bf16 split of the fp32 contraction
    - kernel-level parity of real v_mfma_f32_16x16x32_bf16 products against fp64 and the loop-level rate against the
    fp32 loop (tools/split_bf16_parity.py, a child process)"""
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "tools", "split_bf16_parity.py"), "--json"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(last[-1]) if last else dict(error=f"rc={r.returncode}", stderr=r.stderr[-300:])
    except Exception as e:  # noqa: BLE001
        return dict(error=f"{type(e).__name__}: {e}"[:300])


def emulated_shards(args):
    """EXTRA, one GPU: rank 0's share of a K-way row sharding (``--emulate-shard K``: the rank's rows, its own range of
    the KL terms, no all-reduce) for K = 2, 4, 8 - a child process each.  Not a scaling measurement (no second device,
    no collective): it is the per-rank step time a K-GPU run would have BEFORE its all-reduce, i.e. an upper bound on
    what the sharding can give."""
    import subprocess

    out = {}
    for k in (2, 4, 8):
        cmd = [sys.executable, os.path.abspath(__file__), "--headline-only", "--blocks", "3", "--emulate-shard", str(k),
               "--steps", str(args.steps), "--warmup", str(args.warmup)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(last[-1]) if last else None
            out[str(k)] = dict(ms_per_step=d["ms_per_step"]) if d else dict(error=f"rc={r.returncode}")
        except Exception as e:  # noqa: BLE001
            out[str(k)] = dict(error=f"{type(e).__name__}: {e}"[:200])
    out["note"] = ("rank 0's share of a K-way row sharding on ONE GPU, no all-reduce: per-rank step time before the "
                   "collective, not a scaling measurement")
    return out


def extra_workload(which, args):
    """the JSON line of ``bench.py --workload which`` (a child process: nothing it does can take the contract line down),
    cut to what the default line carries for it"""
    import subprocess

    def with_cache(cmd):  # config 1 once more with the engine's hipGraph cache on (off by default: csrc/step.hip)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, GPSA_STEP_GRAPH="1"))
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            return dict(steps_per_s=d["value"], verbatim_loop_steps_per_s=(d.get("verbatim_loop") or {}).get("value"),
                        **(d.get("step_graph_cache") or {}))
        except Exception as e:
            return dict(error=f"{type(e).__name__}: {e}"[:200])

    # (config 1's step is ~0.6 ms of ~60 tiny launches: the first ~1000 steps of a process run 30-70 % slow on these
    #  boxes (clocks), so its child warms up for 1000 steps and times blocks of at least 200)
    steps, warm = (max(args.steps, 200), max(args.warmup, 1000)) if which == "1" else (args.steps, args.warmup)
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", which, "--steps", str(steps), "--warmup",
           str(warm), "--blocks", "3"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200)
        last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not last:
            return dict(error=f"rc={r.returncode}", stderr=r.stderr[-300:])
        d = json.loads(last[-1])
    except Exception as e:
        return dict(error=f"{type(e).__name__}: {e}"[:300])
    out = dict(workload=d["config"]["workload"], eager_steps_per_s=d["value"], ms_per_step=d["ms_per_step"],
               ms_per_step_min_max=[d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"]])
    if d.get("graph_replay"):
        out["graph_replay_steps_per_s"] = d["graph_replay"].get("value")
    if which == "1":
        out["with_step_graph_cache"] = with_cache(cmd)
    if d.get("verbatim_loop"):
        out["verbatim_loop"] = {k: d["verbatim_loop"][k] for k in ("value", "ms_per_step", "over_headline_loop", "loop")}
    if d.get("cpu_baseline"):
        cb = d["cpu_baseline"]
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample")}
        out["gpu_over_cpu"] = d["value"] / cb["value"]
    roof = d.get("roofline")
    if roof:
        out["contractions"] = {"forward": dict(kernel=roof["kernel"].split(" (")[0], avg_ms=roof["avg_launch_ms"],
                                               frac=roof["frac"], executed_frac=roof["executed_frac"]),
                               **{k: {kk: v.get(kk) for kk in ("kernel", "avg_ms", "frac", "nominal_frac") if kk in v}
                                  for k, v in roof.get("other_kernels", {}).items()}}
    return out


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher around it: start the N ranks ourselves (torchrun's module
    as a CHILD process, before this process has touched the GPU), let rank 0's JSON line through on our stdout and
    leave with the children's status."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's only working mode on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    # the children's stdout is relayed line by line: JSON lines to our stdout, anything else (gloo's connection
    # chatter, launcher notices) to stderr, so that stdout stays "ONE JSON line"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for ln in proc.stdout:
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln)
        sys.stdout.flush()
    return proc.wait()


_SMI_SAMPLER = r'''
import subprocess, sys, time, os
out, stop, life = sys.argv[1], sys.argv[2], float(sys.argv[3])
t_end = time.time() + life
go = stop + ".go"
while time.time() < t_end and not os.path.exists(go) and not os.path.exists(stop):
    time.sleep(0.05)  # idle until the sustained run begins: no rocm-smi call lands inside the timed blocks
with open(out, "w") as f:
    while time.time() < t_end and not os.path.exists(stop):
        t = time.time()
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        except Exception as e:
            r = "rocm-smi failed: %r" % (e,)
        keep = [ln.strip() for ln in r.splitlines() if ("Power" in ln or "sclk" in ln or "failed" in ln)]
        f.write("%.3f\t%s\n" % (t, " | ".join(keep)))
        f.flush()
        time.sleep(max(0.0, 0.25 - (time.time() - t)))
'''


def start_smi_sampler(life_s=900.0):
    """a child process that asks rocm-smi for the package power and the shader clock four times a second, from the
    moment the sustained run begins (a ``.go`` file) until told to stop; started BEFORE this process touches the GPU (it
    never does so itself: rocm-smi reads sysfs) and idle until then"""
    import subprocess
    import tempfile

    d = tempfile.mkdtemp(prefix="gpsa_smi_")
    out, stop = os.path.join(d, "samples.tsv"), os.path.join(d, "stop")
    try:
        proc = subprocess.Popen([sys.executable, "-c", _SMI_SAMPLER, out, stop, str(life_s)],
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:  # noqa: BLE001
        return None
    return dict(proc=proc, out=out, stop=stop)


def stop_smi_sampler(smp, t0, t1):
    """-> what rocm-smi reported between t0 and t1 (time.time() stamps): samples, shader clock (MHz) and power (W)"""
    import re

    if smp is None:
        return dict(error="sampler not started")
    try:
        open(smp["stop"], "w").close()
        smp["proc"].wait(timeout=30)
    except Exception:  # noqa: BLE001
        pass
    clk, pw, n, raw = [], [], 0, None
    try:
        for ln in open(smp["out"]):
            ts, _, txt = ln.partition("\t")
            if not (t0 <= float(ts) <= t1):
                continue
            n += 1
            raw = txt.strip()[:200]
            c = re.search(r"sclk[^|]*\((\d+)Mhz\)", txt)
            w = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
            if c:
                clk.append(int(c.group(1)))
            if w:
                pw.append(float(w.group(1)))
    except OSError as e:
        return dict(error=str(e))
    st = lambda v: dict(min=min(v), mean=round(sum(v) / len(v), 1), max=max(v)) if v else None  # noqa: E731
    return dict(samples=n, sclk_mhz=st(clk), power_w=st(pw), last_sample=raw,
                source="rocm-smi --showpower --showclocks from a child process started before the first GPU call, "
                       "4 samples/s, those inside the sustained run's wall-clock window")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    want_sustained = (args.sustained > 0 and args.workload == "2" and not args.no_extras and not args.graph_only
                      and args.emulate_shard <= 1)
    smi = start_smi_sampler() if (want_sustained and rank == 0) else None
    # diagnostics for a 1-GPU box: GPSA_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and the collectives
    # on gloo (RCCL refuses two ranks per device), which exercises this file's N > 1 path end to end
    one_dev = os.environ.get("GPSA_BENCH_ONE_DEVICE", "0") == "1"
    if one_dev:
        local = 0
    dev = torch.device(f"cuda:{local}")
    rank_devices = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # first contact with RCCL: any failure - the rendezvous, the communicator, the first collective - ends THIS
        # rank with the library's own message and a non-zero status (the launcher then takes the others down); nothing
        # is retried and no other backend is substituted
        try:
            torch.cuda.set_device(dev)
            if one_dev:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            probe = torch.full((1,), float(rank + 1), device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if abs(float(probe.item()) - world * (world + 1) / 2) > 1e-6:
                raise RuntimeError(f"first all-reduce over {world} ranks returned {float(probe.item())}")
            names = [None] * world
            dist.all_gather_object(names, f"rank {rank}: cuda:{local} ({torch.cuda.get_device_name(dev)})")
            rank_devices = names
        except Exception as e:
            sys.stderr.write(f"bench.py: rank {rank} of {world}: {'gloo' if one_dev else 'RCCL'} first contact failed on "
                             f"cuda:{local}: {type(e).__name__}: {e}\n")
            sys.stderr.flush()
            os._exit(3)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    torch.cuda.set_device(dev)

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    from spatial_alignment_amd import ops as ops_mod
    from spatial_alignment_amd.parallel import GradAllReducer, shard_data_dict
    from spatial_alignment_amd.synthetic import make_grid_problem, make_model

    from spatial_alignment_amd.parallel import setup_output_sharding, shard_outputs

    dd_full = make_grid_problem(side=args.side, n_views=args.views, n_outputs=args.outputs, device="cpu",
                                compute_device=dev if args.workload != "2" else None)
    emu = max(1, args.emulate_shard) if world == 1 else 1
    by_outputs = args.shard == "outputs" and world * emu > 1
    import spatial_alignment_amd as gp

    mkw = dict(m=args.M, device=dev, fixed_view_idx=args.fixed)
    if args.latent is not None:
        mkw["n_latent_gps"] = {"expression": args.latent}
    if args.warp == "matern12":
        mkw["kernel_func_warp"] = gp.matern12_kernel
    if by_outputs:  # every row, this rank's slice of the outputs; the model is built on the slice
        dd = shard_outputs(dd_full, rank, world * emu)
        model = make_model(dd, **mkw)
    else:
        model = make_model(dd_full, **mkw)  # identical on every rank (seeded)
        dd = shard_data_dict(dd_full, rank, world * emu)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if args.workload in ("1", "2") else {}
    dd = {m: {"spatial_coords": d["spatial_coords"].to(dev), "outputs": d["outputs"].to(dev),
              "n_samples_list": d["n_samples_list"]} for m, d in dd.items()}
    out_reducer = None
    if by_outputs:
        out_reducer = setup_output_sharding(model, rank, world * emu)  # (broadcasts the shared parameters)
    elif args.kl_share == "owner" and world * emu > 1:
        from spatial_alignment_amd.parallel import own_kl_terms

        own_kl_terms(model, rank, world * emu)  # a contiguous range of the KL terms at weight 1 (owner computes)
    else:
        model.kl_scale = 1.0 / (world * emu)
    if args.no_check:
        model.check_numerics = False
    if args.overlap:
        model.overlap_views = True
    view_idx, Ns, _, _ = model.create_view_idx_dict(dd)
    Xs = {m: d["spatial_coords"] for m, d in dd.items()}
    torch.manual_seed(1000 + rank)
    if args.graph_only:
        from spatial_alignment_amd.train import GraphedTrainStep

        from spatial_alignment_amd.optim import FusedAdam

        gopt = FusedAdam(model.parameters(), lr=1e-2)  # device-side step counter: capturable
        gs = GraphedTrainStep(model, gopt, dd, view_idx, Ns, S=args.S, warmup=3)
        for _ in range(2):
            gs.step()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        for _ in range(args.steps):
            gs.step()
        torch.cuda.synchronize()
        gdt = time.perf_counter() - g0
        gs.check()
        print(json.dumps(dict(value=args.steps / gdt, unit="steps/s", ms_per_step=1e3 * gdt / args.steps,
                              final_loss=float(gs.loss.item()),
                              note="same step (forward+ELBO+backward+Adam) as ONE hipGraph replay")),
              flush=True)
        return
    from spatial_alignment_amd.optim import FusedAdam
    from spatial_alignment_amd.train import train_step

    opt = FusedAdam(model.parameters(), lr=1e-2)  # torch.optim.Adam's update as one HIP launch
    reducer = out_reducer if out_reducer is not None else GradAllReducer(
        model.parameters(), overlap=args.overlap_reduce and world > 1, model=model)
    timer = KernelTimer(model, args.steps * max(1, args.blocks))
    timer.S = args.S

    def reference_step(S):
        # the reference's two calls and its loop order (examples/grid_example.py:62-78: forward, loss_fn, zero_grad,
        # backward, optimiser step; + the all-reduce of the gradient on N > 1) with this package's FusedAdam
        # (torch.optim.Adam's update as one launch) and WITHOUT the reference's per-step loss.item(); the loop exactly
        # as the reference writes it is ``verbatim_step`` below ("verbatim_loop" in the line).  Nothing tells forward
        # what loss_fn will get: forward leaves the data GP to loss_fn, which runs it with the likelihood folded in
        G_means, G_samples, F_latent_samples, F_samples = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=S)
        loss = model.loss_fn(dd, F_samples)
        opt.zero_grad()
        loss.backward()
        reducer()
        opt.step()
        return loss

    vopt = [None]

    def verbatim_step(S):
        # examples/grid_example.py:59-78 VERBATIM: torch.optim.Adam(model.parameters(), lr=1e-2), and the host reads
        # the loss every step (loss.item(), :78) - what a user who points the script at this package gets
        if vopt[0] is None:
            vopt[0] = torch.optim.Adam(model.parameters(), lr=1e-2)
        optimizer = vopt[0]
        G_means, G_samples, F_latent_samples, F_samples = model.forward(X_spatial=Xs, view_idx=view_idx, Ns=Ns, S=S)
        loss = model.loss_fn(dd, F_samples)
        optimizer.zero_grad()
        loss.backward()
        reducer()
        optimizer.step()
        loss.item()
        return loss

    def helper_step(S):  # the same loop body as the package's helper (seed gradient kept on the device)
        return train_step(model, opt, dd, view_idx, Ns, S=S, reducer=reducer, static_grads=args.static_grads)

    def time_blocks(step, S, blocks, before=None):
        """``blocks`` x (barrier + synchronize, --steps steps, synchronize + barrier), MAX over ranks per block"""
        for _ in range(args.warmup):
            step(S)
        # host hygiene, once per process: the construction garbage of torch and the model (~74k objects) otherwise
        # meets its one full collection somewhere in the timed blocks (0.05 - 0.15 s: one block at 10 - 15 ms per
        # step in every run of rounds 5 and 6; the median never saw it).  The collector stays ON.
        if not _GC_DONE:
            import gc
            gc.collect()
            gc.freeze()
            _GC_DONE.append(True)
        out, loss = [], None
        for b in range(blocks):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            if before is not None and b == 0:
                before()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step(S)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            out.append(float(t.item()))
        return out, loss

    def summary(ts):
        srt = sorted(ts)
        med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        return dict(value=args.steps / med, unit="steps/s", ms_per_step=1e3 * med / args.steps, blocks=len(ts),
                    steps_per_block=args.steps, ms_per_step_min=1e3 * srt[0] / args.steps,
                    ms_per_step_max=1e3 * srt[-1] / args.steps,
                    ms_per_step_blocks=[round(1e3 * t / args.steps, 4) for t in ts])

    nblk = max(1, args.blocks)
    # (with the engine's hipGraph cache asked for - GPSA_STEP_GRAPH=1, config 1's extra leg - no kernel-timing events: the
    #  engine does not replay with them on)
    cache_on = os.environ.get("GPSA_STEP_GRAPH") == "1"
    times, loss = time_blocks(reference_step, args.S, nblk, before=None if cache_on else timer.start)
    head = summary(times)
    dt = args.steps / head["value"]  # the median block
    final_loss = float(loss.item())
    # the contraction kernels' HIP-event timings of exactly these blocks (read now: the later legs run the same plan)
    ks_head = None
    if rank == 0:
        ks_head = timer.summary(args.M, args.S * int(sum(dd["expression"]["n_samples_list"])),  # this rank's columns
                                int(args.latent or dd["expression"]["outputs"].shape[1]))       # and (latent) outputs
    # the package's own helper around the same two calls (what round 3's line timed): same kernels, one launch less
    helper = summary(time_blocks(helper_step, args.S, min(nblk, 3))[0]) if (args.workload == "2" and
                                                                            not args.headline_only) else None
    # the reference's loop verbatim (torch.optim.Adam + a host read of the loss every step)
    verbatim = None
    if args.workload in ("1", "2") and not args.headline_only:
        vb = summary(time_blocks(verbatim_step, args.S, min(nblk, 3))[0])
        verbatim = dict(value=vb["value"], ms_per_step=vb["ms_per_step"],
                        ms_per_step_min_max=[vb["ms_per_step_min"], vb["ms_per_step_max"]],
                        over_headline_loop=vb["ms_per_step"] / head["ms_per_step"],
                        loop="examples/grid_example.py:59-78 verbatim: torch.optim.Adam(model.parameters(), lr=1e-2); "
                             "forward; loss_fn; optimizer.zero_grad(); loss.backward(); optimizer.step(); loss.item()")
    # the same loop with the inducing-point gradient in the other mode (model.exact_inducing_grad)
    exact_info = None
    if args.workload == "2" and not args.headline_only:
        timed_exact = bool(model._cache.plan.exact)  # (the plan the last timed step ran)
        saved_mode = model.exact_inducing_grad
        model.exact_inducing_grad = not timed_exact
        try:
            other = summary(time_blocks(reference_step, args.S, min(nblk, 3))[0])
        finally:
            model.exact_inducing_grad = saved_mode
        on, off = (head, other) if timed_exact else (other, head)
        exact_info = dict(timed=timed_exact, ms_per_step_on=on["ms_per_step"], ms_per_step_off=off["ms_per_step"],
                          note="on: every gradient within 1e-4 of the reference's fp64 run (what the parity tests and "
                               "parity_at_bench_size hold the step to); off: grad Gtilde ~7e-4 at M = 200, everything "
                               "else unchanged")

    # secondary: the same step at S = 1, the reference's forward default; same protocol
    s1 = None
    if args.S != 1 and not args.no_s1:
        s1 = dict(S=1, **summary(time_blocks(reference_step, 1, min(nblk, 3))[0]))

    # sustained: --sustained consecutive headline steps (1000 = ~7 s on one GPU), one synchronise per 100, next to what
    # rocm-smi saw of the clock and the power meanwhile - the timed blocks above are 5 x 20 steps
    sustained = None
    if want_sustained:
        nb_s = max(1, args.sustained // 100)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if smi is not None:
            open(smi["stop"] + ".go", "w").close()
        w0 = time.time()
        tb = []
        for b in range(nb_s):
            t0 = time.perf_counter()
            for _ in range(100):
                reference_step(args.S)
            torch.cuda.synchronize()
            tb.append(time.perf_counter() - t0)
        if world > 1:
            dist.barrier()
        w1 = time.time()
        tot = torch.tensor([sum(tb)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.MAX)
        sustained = dict(steps=100 * nb_s, seconds=round(float(tot.item()), 3), value=100 * nb_s / float(tot.item()),
                         unit="steps/s", ms_per_step=1e3 * float(tot.item()) / (100 * nb_s),
                         ms_per_step_min_block=round(1e3 * min(tb) / 100, 4),
                         ms_per_step_max_block=round(1e3 * max(tb) / 100, 4), steps_per_block=100,
                         note="the headline step (the reference's two calls + FusedAdam), back to back; rank 0's blocks")
        if rank == 0:
            sustained["device"] = stop_smi_sampler(smi, w0, w1)
    elif smi is not None:
        stop_smi_sampler(smi, 0, 0)

    graph_info = None
    if world == 1 and not args.no_graph and not args.graph_only:
        # extra: the SAME step captured into a hipGraph and replayed (train.GraphedTrainStep); run in a
        # child process so that nothing it does can take the contract line down with it
        import subprocess

        cmd = [sys.executable, os.path.abspath(__file__), "--graph-only", "--steps", str(args.steps),
               "--S", str(args.S), "--emulate-shard", str(emu)] + (
                   ["--workload", args.workload] if args.workload != "2" else
                   ["--side", str(args.side), "--views", str(args.views), "--outputs", str(args.outputs), "--M", str(args.M)])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            graph_info = json.loads(last[-1]) if last else dict(error=f"rc={r.returncode}", stderr=r.stderr[-300:])
        except Exception as e:
            graph_info = dict(error=f"{type(e).__name__}: {e}"[:300])

    if rank == 0:
        N = int(sum(dd_full["expression"]["n_samples_list"]))
        ks = ks_head
        pmc, pmc_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        default_cfg = (args.S, args.side, args.views, args.outputs, args.M, world) == (5, 100, 2, 50, 200, 1)
        if world == 1 and emu == 1 and args.workload == "2" and not args.headline_only and not args.no_traffic:
            live = measure_traffic(args)
            if live is not None and "error" not in live:
                pmc = live
                pmc_source = (f"measured in this run: two child passes of the headline loop under rocprofv3 --pmc "
                              f"(FETCH_SIZE, WRITE_SIZE), {live.get('seconds')} s")
            elif live is not None:
                pmc_source = f"live measurement failed ({live.get('error')}); "
        if pmc is None and os.path.exists(pmc_path) and default_cfg:  # the counters were collected on the default workload
            try:
                pmc = json.load(open(pmc_path))
                pmc_source = (pmc_source or "") + ("profiles/pmc_traffic.json (builder-collected rocprofv3 --pmc passes "
                                                   "on this workload, not measured in this run)")
            except Exception:
                pmc = None
        # flops a launch EXECUTES relative to the nominal 2*C*L*M^2: the forward form and the Gram kernel walk
        # one triangle of the 16x16 tiles (MB tiles per side) and skip all-padding K steps; the accumulate
        # panel pads the M rows up to a multiple of 16
        MB = -(-args.M // 16)
        tri = (MB * (MB + 1) / 2) / (MB * MB)
        pad_k = args.M / (16.0 * MB)
        kept = any(p.saved_bytes > p.saved_bytes_nokeep for p in timer.plans)
        keep_bytes = sum(p.saved_bytes - p.saved_bytes_nokeep for p in timer.plans) if kept else 0
        if args.M <= 256:
            executed_ratio = {
                "quadform_fwd": tri * (16.0 * MB / args.M) ** 2 * pad_k,
                "quadform_bwd_alpha": (16.0 * MB / args.M),
                "quadform_bwd_omega": tri * (16.0 * MB / args.M) ** 2,
            }
            kernel_of = {
                "quadform_fwd": "quad_sym_mfma_kernel (gpsa_quadform_fwd; upper-triangle tiles, all-padding K steps "
                                "skipped: executes 0.54x of the nominal 2*C*L*M^2 flops)",
                "quadform_bwd_alpha": "panel_mfma_kernel<MODE_ACCUM> (gpsa_quadform_bwd_alpha)",
                "quadform_bwd_omega": "gram_mfma_kernel (gpsa_quadform_bwd_omega; lower-triangle tiles: executes "
                                      "0.58x of the nominal flops)",
            }
            # training keeps the data GP's products Omega_l alpha (step engine, io.keep_products): the forward is then
            # the FULL product (rows and contraction padded to 16 MB), the alpha-gradient one streaming read of them
            if kept:
                executed_ratio["quadform_fwd"] = (16.0 * MB / args.M) ** 2
                kernel_of["quadform_fwd"] = ("panel_mfma_kernel<MODE_QUAD> with kept products (gpsa_quadform_fwd_keep_f32: "
                                             "the full 2*C*L*M^2 product, stored once for the backward)")
                kernel_of["quadform_bwd_alpha"] = ("kept_wsum_kernel (gpsa_quadform_bwd_alpha_kept_f32: one streaming "
                                                   "read of the kept products; HBM-bound, no matrix-core work)")
        else:  # the 128 x 128 LDS-DMA kernels: row blocks of 128, contraction padded to 16
            Mp, nrb, M2 = 16 * MB, -(-args.M // 128), float(args.M) ** 2
            executed_ratio = {
                "quadform_fwd": sum(128 * (Mp - 128 * rb) for rb in range(nrb)) / M2,
                "quadform_bwd_alpha": (16.0 * MB / args.M) if MB <= 32 else nrb * 128 * Mp / M2,
                "quadform_bwd_omega": (nrb * (nrb + 1) // 2) * 128 * 128 / M2,
            }
            kernel_of = {
                "quadform_fwd": "big_quad_kernel<TRI> (gpsa_quadform_fwd; block-triangular, form closed in the kernel)",
                "quadform_bwd_alpha": ("panel_mfma_kernel<MODE_ACCUM>" if MB <= 32 else "big_accum_kernel")
                                      + " (gpsa_quadform_bwd_alpha)",
                "quadform_bwd_omega": "gram_big_kernel (gpsa_quadform_bwd_omega; lower-triangle 128 x 128 block pairs)",
            }
            if kept:
                executed_ratio["quadform_fwd"] = nrb * 128 * Mp / M2
                kernel_of["quadform_fwd"] = ("big_quad_kernel<STORE> with kept products (gpsa_quadform_fwd_keep_f32: the "
                                             "full product, stored once, form closed in the kernel)")
                kernel_of["quadform_bwd_alpha"] = ("col_wsum_rows_kernel (gpsa_quadform_bwd_alpha_kept_f32: one streaming "
                                                   "read of the kept products; HBM-bound)")
        fused = bool(getattr(model.__dict__.get("_cache"), "fuse", None))
        if fused and args.M <= 208:
            # the training helpers fold the likelihood into the data GP's pass (gpsa_quadform_elbo_f32): the full
            # product, closed and weighted into the alpha-gradient in the accumulators - nothing kept, nothing re-read
            # rows padded to 16 MB; the contraction index is NOT padded: the all-padding K steps of the last chunk are
            # skipped (RL = 2 of 4 when M - 16 (MB - 1) <= 8)
            k_eff = 16 * (MB - 1) + 4 * (2 if args.M - 16 * (MB - 1) <= 8 else 4)
            executed_ratio["quadform_fwd"] = (16.0 * MB / args.M) * (k_eff / float(args.M))
            kernel_of["quadform_fwd"] = ("panel_elbo_kernel (gpsa_quadform_elbo_f32: the full 2*C*L*M^2 product, with the "
                                         "variance, draw, Gaussian likelihood, its gradient and abar = 2 sum_l g_l Omega_l "
                                         "alpha formed from the accumulators)")
            kernel_of["quadform_bwd_alpha"] = ("what is left of the alpha-gradient: elbo_post_kernel (column sums of g) + "
                                               "the mean term's [M,L] x [L,C] product; no pass over the products")
            kept = False
        roof = None
        if ks:
            # the dominant kernel = the contraction with the longest launch
            dname = max(ks, key=lambda k: ks[k]["avg_ms"])
            dom = ks[dname]
            ex = lambda k: ks[k]["tflops"] * executed_ratio[k]
            roof = dict(bound="mfma", kernel=kernel_of[dname], achieved=dom["tflops"],
                        peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=dom["tflops"] / PEAK_F32_MFMA_TFLOPS,
                        executed_frac=ex(dname) / PEAK_F32_MFMA_TFLOPS,
                        traffic=((pmc or {}).get("hbm_bytes_per_launch") or {}).get(dname),
                        traffic_source=pmc_source,
                        traffic_note=(pmc or {}).get("note"),
                        avg_launch_ms=dom["avg_ms"], flops_per_launch=dom["flops"],
                        flops_note="achieved / frac: algorithmic 2*C*L*M^2 per launch (C = S*N columns); "
                                   "executed_frac: the flops the kernel issues (padding in, skipped tiles out)",
                        # frac = on EXECUTED flops (what the matrix pipe did); nominal_frac = on the algorithmic
                        # 2*C*L*M^2 (these two kernels skip half the tiles, so it exceeds 1)
                        other_kernels={k: dict(kernel=kernel_of[k], avg_ms=v["avg_ms"],
                                               tflops_executed=ex(k), frac=ex(k) / PEAK_F32_MFMA_TFLOPS,
                                               nominal_tflops=v["tflops"],
                                               nominal_frac=v["tflops"] / PEAK_F32_MFMA_TFLOPS)
                                       for k, v in ks.items() if k != dname})
            if kept and "quadform_bwd_alpha" in roof["other_kernels"]:  # a streaming kernel: bytes, not flops
                o = roof["other_kernels"]["quadform_bwd_alpha"]
                gbs = keep_bytes / (o["avg_ms"] * 1e-3) / 1e9
                roof["other_kernels"]["quadform_bwd_alpha"] = dict(
                    kernel=o["kernel"], avg_ms=o["avg_ms"], bound="hbm", bytes_per_launch=keep_bytes,
                    achieved=gbs, peak=8000.0, unit="GB/s", frac=gbs / 8000.0)
            # step level: what the matrix cores execute per step, against the peak (the SURVEY 8d formula counts three
            # full products per step where this step executes one full product and a triangular Gram: its rate
            # exceeded the peak in round 3's record and said nothing - it is given as a flop count only)
            Mq, Cq, Lq = args.M, args.S * N, (args.latent or args.outputs)
            step_flops = 3.0 * (2.0 * Cq * Mq * Mq * (Lq + 1) + 2.0 * Cq * Mq * Lq
                                + sum(2.0 * n_v * Mq * Mq * 3 for n_v in dd_full["expression"]["n_samples_list"]))
            roof["step_level"] = dict(reference_algorithm_flops_per_step=step_flops,
                                      note="reference_algorithm_flops_per_step: 3 x forward flops of the reference's "
                                           "step (SURVEY 8d), for scale only; executed_*: the contraction flops this "
                                           "step's kernels issue / median step time / peak, per GPU")
            # what the matrix cores actually issue for the three contractions of this step (with kept products the
            # alpha-gradient re-uses the forward's product: no flops)
            exec_fl = sum(ks[k]["flops"] * executed_ratio[k] for k in ks
                          if not ((kept or fused) and k == "quadform_bwd_alpha"))
            if fused and "quadform_bwd_alpha" in roof["other_kernels"]:
                o = roof["other_kernels"]["quadform_bwd_alpha"]
                roof["other_kernels"]["quadform_bwd_alpha"] = dict(kernel=o["kernel"], avg_ms=o["avg_ms"])
            roof["fused_elbo"] = fused
            if fused:
                roof["note"] = ("this launch also carries the Gaussian likelihood and the alpha-gradient - the work of "
                                "panel_mfma_kernel<QUAD> (3.25-3.31 ms, frac 0.77 / executed 0.84), kept_wsum_kernel "
                                "(0.68-0.81 ms, 4.2 GB read) and four elementwise kernels (0.08 ms) of the unfused step "
                                "(GPSA_FUSE_ELBO=0): frac counts the product's 2*C*L*M^2 flops alone")
            roof["step_level"]["executed_contraction_flops_per_step"] = exec_fl
            roof["step_level"]["executed_contraction_tflops"] = exec_fl * (args.steps / dt) / 1e12
            roof["step_level"]["executed_contraction_frac"] = exec_fl * (args.steps / dt) / 1e12 / PEAK_F32_MFMA_TFLOPS
        line = {
            "metric": "training steps/sec (ELBO fwd+bwd), 2-view N=10k M=200, 1/2/4/8 GPU" if args.workload == "2" else
                      f"training steps/sec (ELBO fwd+bwd), BASELINE config {args.workload} at its stated size (diagnostic)",
            "value": args.steps / dt,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "timing": {"protocol": f"{head['blocks']} blocks of {args.steps} steps, barrier + synchronize around each, max "
                                   "over ranks per block; value / ms_per_step = the MEDIAN block; one gc.collect() + "
                                   "gc.freeze() after the warm-up steps (the collector stays on)",
                       **{k: head[k] for k in ("blocks", "ms_per_step_min", "ms_per_step_max", "ms_per_step_blocks")}},
            # the timed loop is the reference's two calls in the reference's order with FusedAdam and no per-step host
            # read of the loss (bench.py reference_step); "verbatim_loop": examples/grid_example.py:59-78 exactly
            # (torch.optim.Adam + loss.item()); "helper_loop": the same step through train.train_step
            "reference_loop": {"is_the_headline": True, "value": head["value"], "ms_per_step": head["ms_per_step"],
                               "calls": "model.forward(X_spatial=, view_idx=, Ns=, S=); model.loss_fn(data_dict, F_samples); "
                                        "optimizer.zero_grad(); loss.backward(); optimizer.step()"},
            "helper_loop": helper,
            "verbatim_loop": verbatim,
            "exact_inducing_grad": exact_info,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (fp64 warp layer, covariance whitening and M x M factorisations)",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic 2D grid, {args.views} views x {args.side * args.side} spots, "
                            f"{args.outputs} outputs" + (f" through {args.latent} latent GPs" if args.latent else "")
                            + f", M_G=M_X={args.M}, " + ("Matern-1/2 warp / RBF data" if args.warp == "matern12" else
                                                        "RBF warp+data")
                            + (f", fixed_view_idx={args.fixed}" if args.fixed is not None else "")
                            + f", S={args.S}, forward+ELBO+backward+Adam",
                "n_spots_total": N,
                **({"world_size": dist.get_world_size(), "rank_devices": rank_devices,
                    "overlapped_reduce": bool(args.overlap_reduce)} if world > 1 else {}),
                "parallelism": ("single GPU" if world == 1 else
                                f"outputs sharded x{world}, 1 all-reduce/step of the shared parameters' gradients" if by_outputs
                                else f"rows-of-views sharded x{world}, 1 all-reduce/step"),
                "check_numerics_sync": not args.no_check,
                **({"emulated_shard": f"rank 0 of {emu} (diagnostic, not the contract metric)"} if emu > 1 else {}),
                "final_loss": final_loss,
            },
            "roofline": roof,
            "secondary_S1": s1,
            "sustained": sustained,
            "graph_replay": graph_info,
            "step_graph_cache": step_graph_stats(model),
        }
        # (the CPU leg runs at N = 1 only: at N > 1 the other ranks would sit in the process group's teardown for the
        #  ~45 s rank 0 spends on the host, and the figure would not differ)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, state, dd_full)
            if emu == 1 and world == 1 and args.workload == "2":
                try:
                    line["parity_at_bench_size"] = parity_at_bench_size(args, model, dd_full, dd, view_idx, Ns)
                except Exception as e:  # the contract line must not die with its extra
                    line["parity_at_bench_size"] = dict(error=f"{type(e).__name__}: {e}"[:300])
        else:
            line["cpu_baseline"] = None
        if world == 1 and emu == 1 and not args.no_extras:
            # BASELINE configs 1 and 3 on this GPU, each as a child process running this file with --workload
            line["config1"] = extra_workload("1", args)
            line["config3"] = extra_workload("3", args)
            line["split_bf16_experiment"] = split_bf16_experiment()
            es = emulated_shards(args)
            for k in ("2", "4", "8"):
                if "ms_per_step" in es.get(k, {}):
                    es[k]["step_time_ratio_vs_1"] = line["ms_per_step"] / es[k]["ms_per_step"]
            line["emulated_shards"] = es
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
