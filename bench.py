"""Headline benchmark: forward+backward Mpixels/s of the splat decoder path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): scenes of 65,536 pixel-aligned Gaussians, SH degree 0, rendered at
256x256.  One STEP = one pass of the hot path over one batch that is already resident in HBM:
`--scenes` scenes x `--views` target views per GPU (default 8 x 4 = 32 renders -- BASELINE configs[3]'s per-GPU
share), i.e. decoder forward (projection, tile binning, depth sort, compositing) + MSE loss + full backward to
every Gaussian parameter and to the camera poses.  Renders are independent, so N GPUs shard scene-first with no
data-path collective (weak scaling: every rank gets its own 8 x 4 batch).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel's algorithmic HBM bytes / its HIP-event duration over the timed region
  cpu_baseline  the CPU oracle (oracle/splat_ref.py, kind "port") timed on the host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0           # measured float4 copy ceiling


def stage_bytes(stage: str, S: int, V: int, G: int, K: int, P: int, D_total: int) -> float:
    """Algorithmic (compulsory) HBM bytes of ONE launch of a stage: SURVEY.md 8(d) per-unit figures x units."""
    R = S * V
    if stage == "project_fwd":
        return S * G * (44 + 12 * K) + R * G * 48.0          # scene parameters once, one record per render
    if stage == "bin_pairs":
        return R * G * 8.0 + 12.0 * D_total
    if stage == "tile_sort":
        return 24.0 * D_total
    if stage == "render_fwd":
        return 40.0 * D_total + 28.0 * R * P
    if stage == "render_bwd":
        return 40.0 * D_total + 32.0 * R * P + 40.0 * R * G
    if stage == "project_bwd":
        return S * G * (44 + 12 * K) * 2.0 + 88.0 * R * G
    return 0.0


def total_bytes(S, V, G, K, P, D_total) -> float:
    """Whole fwd+bwd path per step: A = G(308+36K) + 124 D + 60 P per render."""
    return S * V * (G * (308 + 36 * K) + 60.0 * P) + 124.0 * D_total


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(args, batch_cpu) -> dict:
    """CPU oracle, fwd+bwd, same workload, bounded sample (first scene, one view at a time until ~`--cpu-budget`
    seconds are spent).  The oracle's per-tile tensors are small, so more than ~16 threads only adds
    synchronisation cost: threads = min(host cores, 16), and that is the `cores` reported."""
    from tests import util
    from spfsplatv2_amd import synthetic as syn
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    S, V = batch_cpu.extrinsics.shape[:2]
    h, w = batch_cpu.image_shape
    done, spent = 0, 0.0
    while done < S * V and (done == 0 or spent + spent / done < args.cpu_budget):
        si, vi = divmod(done, V)
        sub = syn.Batch(**{k: (t[si:si + 1, vi:vi + 1] if k in ("extrinsics", "intrinsics", "near", "far", "target")
                               else (t[si:si + 1] if isinstance(t, torch.Tensor) else t))
                           for k, t in batch_cpu.__dict__.items()})
        t0 = time.perf_counter()
        util.run_oracle(sub, torch.float32, want_fragile=False)
        spent += time.perf_counter() - t0
        done += 1
        log(f"cpu_baseline: {done} render(s), {spent:.1f} s")
    return {"value": round(done * h * w / spent / 1e6, 5), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"{done} of the step's {S * V} renders, same workload "
                      f"({batch_cpu.means.shape[1]} Gaussians, {h}x{w}), oracle/splat_ref.py fwd+bwd in float32, "
                      f"{spent:.1f} s on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes", type=int, default=8, help="scenes per GPU per step")
    ap.add_argument("--views", type=int, default=4, help="target views per scene")
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C5"])
    ap.add_argument("--s-mult", type=float, default=1.0)
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU oracle time to spend")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact", action="store_true",
                    help="size the pair buffer from a 16-byte device->host read-back in every step (one host sync per "
                         "step) instead of the default: a PairBudget planned from the first step, verified on the "
                         "device, checked once after the timed region")
    ap.add_argument("--sync-free", action="store_true", help="(default now; kept for compatibility)")
    ap.add_argument("--torch-loss", action="store_true",
                    help="photometric MSE through torch.nn.functional.mse_loss (five eager kernels) instead of the "
                         "fused spfsplatv2_amd.mse_loss (LossMse, loss_mse.py:36-51)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only to exercise the multi-rank path on a "
                         "single-GPU box together with --one-device)")
    ap.add_argument("--one-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--graph", action="store_true",
                    help="capture one whole step (decoder fwd + loss + bwd) in a HIP graph and replay it "
                         "(implies --sync-free)")
    ap.add_argument("--allreduce", action="store_true",
                    help="outer-training-step variant (BASELINE config 5): every rank renders its own views of the "
                         "SAME scenes and the Gaussian-parameter gradients are summed with one RCCL all-reduce")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    builder = rank == 0 if args.one_device else local_rank == 0
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # the in-tree library normally travels with the tree; (re)build it if it is missing or stale (one rank per
    # node compiles, the others wait) -- a no-op when the sources' digest matches
    from spfsplatv2_amd import build as _build
    if builder:
        _build.build(verbose=False)
    if world > 1:
        dist.barrier()
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import _lib, synthetic as syn

    S, V = args.scenes, args.views
    from spfsplatv2_amd import shard
    if args.allreduce:      # same scenes everywhere, rank-specific target poses
        batch_cpu = syn.make_batch(args.config, S, V, seed=1000, s_mult=args.s_mult)
        pg = torch.Generator().manual_seed(7000 + rank)
        batch_cpu.extrinsics = torch.stack([syn.target_poses(pg, V) for _ in range(S)])
    else:                   # independent scenes per rank (scene-first sharding of a larger batch)
        batch_cpu = syn.make_batch(args.config, S, V, seed=1000 + rank, s_mult=args.s_mult)
    b = batch_cpu.to(dev)
    h, w = b.image_shape
    G, K = b.means.shape[1], b.harmonics.shape[-1]
    names = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")
    leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in names}
    bg = torch.zeros(3, device=dev)
    one = torch.ones((), device=dev)
    max_pairs = None

    def step():
        for t in leaves.values():
            t.grad = None
        color, depth, _alpha = spf.render_views(
            leaves["extrinsics"], b.intrinsics, b.near, b.far, (h, w), bg, leaves["means"], leaves["harmonics"],
            leaves["opacities"], leaves["rotations"], leaves["scales"], scale_invariant=True,
            enable_cov_grad=True, enable_sh_grad=True, max_pairs=max_pairs)
        loss = (torch.nn.functional.mse_loss if args.torch_loss else spf.mse_loss)(color, b.target)
        loss.backward(gradient=one)          # (a cached dL/dloss = 1 saves autograd's fill kernel)
        if args.allreduce:
            shard.allreduce_gaussian_grads([leaves[n].grad for n in names[:5]])
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"batch resident: {S} scenes x {V} views, G={G}, K={K}, {h}x{w}")
    step()
    torch.cuda.synchronize(dev)
    D_total = spf.last_forward_stats()["num_pairs"]
    log(f"first step done: D={D_total}, max tile list={spf.last_forward_stats()['max_tile_list']}")
    if not args.exact or args.graph:
        max_pairs = spf.plan_pair_budget(slack=1.25, check="deferred")
        log(f"planned budget: {max_pairs}")
    run = step
    eager_survey = None
    if args.graph:
        # per-stage survey and dominant-kernel timing need eager launches (events are recorded at launch time,
        # a replayed graph launches nothing from the host)
        _lib.stage_timing_enable(True)
        for _ in range(max(args.warmup, 3)):
            step()
        torch.cuda.synchronize(dev)
        eager_survey = {k: (v[0] / v[1], 1) for k, v in _lib.stage_times().items() if k != "rope2d" and v[1] > 0}
        _lib.stage_timing_enable(False)
        for t in leaves.values():
            t.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        run = graph.replay
        log("step captured in a HIP graph")
    # warm-up doubles as the per-stage survey (HIP events around every stage); the timed region then keeps
    # events only around the dominant kernel, so the headline number is not diluted by 14 event records/step
    _lib.stage_timing_enable(True)
    for _ in range(max(args.warmup, 1)):
        run()
    torch.cuda.synchronize(dev)
    survey = {k: v for k, v in _lib.stage_times().items() if k != "rope2d" and v[1] > 0}
    if eager_survey is not None:
        survey = {k: (v[0] * max(args.warmup, 1), max(args.warmup, 1)) for k, v in eager_survey.items()}
    dom = max(survey, key=lambda k: survey[k][0])
    _lib.stage_timing_enable([dom])
    _lib.stage_timing_sample_every(4)     # events around every 4th launch of the dominant kernel
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    stages = _lib.stage_times()
    _lib.stage_timing_enable(False)
    _lib.stage_timing_sample_every(1)
    if max_pairs is not None and spf.last_plan_flags() != 0:
        raise RuntimeError(f"the planned pair budget did not hold (flags {spf.last_plan_flags()}): results invalid")
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        P = h * w
        renders = world * S * V
        value = renders * P * args.steps / dt / 1e6
        dom_ms = (stages[dom][0] / stages[dom][1]) if stages[dom][1] else survey[dom][0] / survey[dom][1]
        dom_bytes = stage_bytes(dom, S, V, G, K, P, D_total)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        prof = ROOT / "profiles" / "pmc_summary.json"
        default_workload = (args.config == "C2" and S == 8 and V == 4 and args.s_mult == 1.0 and not args.allreduce)
        if prof.exists() and default_workload:      # the PMC passes were collected on exactly this workload
            try:
                traffic = json.loads(prof.read_text()).get(_lib.stage_kernel_name(dom), {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        A = total_bytes(S, V, G, K, P, D_total)
        out = {
            "metric": "Mpixels/s fwd+bwd, 256x256 @ ~65k Gaussians" if args.config == "C2" else
                      f"Mpixels/s fwd+bwd ({args.config})",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {G} pixel-aligned Gaussians/scene, SH degree "
                                   f"{int(K ** 0.5) - 1}, {h}x{w}, {S} scenes x {V} views per GPU per step, "
                                   "decoder fwd + MSE + bwd to all Gaussian parameters and poses",
                       "loss": "torch.nn.functional.mse_loss" if args.torch_loss else "spfsplatv2_amd.mse_loss (fused HIP)",
                       "scenes_per_gpu": S, "views_per_scene": V, "gaussians_per_scene": G, "image": [h, w],
                       "sh_coeffs": K, "renders_per_step": renders, "pairs_per_render": round(D_total / (S * V), 1),
                       "s_mult": args.s_mult, "pair_buffer": "exact (read-back per step)" if max_pairs is None else
                                      f"planned from step 0 (x1.25 = {max_pairs.capacity} pairs), verified on device",
                       "launch": "hip-graph replay" if args.graph else "eager",
                       "sharding": ("views of the same scenes per rank + RCCL all-reduce of Gaussian grads"
                                    if args.allreduce else "scene-first, no data-path collective")},
            "roofline": {"bound": "hbm", "kernel": _lib.stage_kernel_name(dom), "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "launch_ms": round(dom_ms, 5),
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "limiter": "VALU issue, not HBM: SQ_ACTIVE_INST_VALU = 78 % of the SIMD cycles in this kernel "
                                    "(profiles/r01_sq_counters.txt)" if dom == "render_bwd" else None,
                         "path_achieved_GBs": round(A / (dt / args.steps) / 1e9, 2),
                         "path_frac_of_copy_ceiling": round(A / (dt / args.steps) / 1e9 / HBM_COPY_GBS, 5)},
            "stage_ms_per_step_warmup": {k: round(v[0] / max(args.warmup, 1), 5) for k, v in survey.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, batch_cpu)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
