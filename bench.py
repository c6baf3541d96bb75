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

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself -- `python -m torch.distributed.run`
on 127.0.0.1, one process per GPU, backend nccl (= RCCL) -- and rank 0 prints the line with `n_gpus: N`.

Launch: one whole step is captured in a HIP graph and replayed (the step's GPU time, ~0.41 ms, is below what one
Python thread needs to launch its kernels one by one); every 10th step of the timed region is the same step launched
eagerly, with HIP events around the dominant kernel -- the roofline's live duration.  `--eager`: kernel-by-kernel
launches for every step (host-bound).

Timing: the `--steps` loop (EXACTLY K steps between barrier + synchronize on both sides, MAX over ranks) is run as
>= 25 back-to-back TRIALS adding up to >= 1 s of GPU work; `ms_per_step` / `value` are the MEDIAN trial and
`trials_ms` keeps every trial (a single 10 ms sample was a 2 % lottery).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel's algorithmic HBM bytes / its HIP-event duration over the timed region, plus
                `valu`: the same kernel's VALU-issue roofline (it is instruction-issue bound, not HBM bound)
  cpu_baseline  the CPU oracle (oracle/splat_ref.py, kind "port") timed on the host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
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
CU_COUNT, SIMD_PER_CU, CLOCK_HZ = 256, 4, 2.4e9   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock

# workload -> (default scenes per GPU, default views per scene); sizes live in spfsplatv2_amd/synthetic.py::CONFIGS
WORKLOADS = {
    "C2": (8, 4),        # BASELINE configs[1] at configs[3]'s per-GPU batch: the headline metric
    "C3": (2, 4),        # 320k Gaussians
    "C5": (1, 8),        # 500k Gaussians, SH degree 3, 512x512
    "REF2V": (16, 1),    # what the shipped 2-view model really renders: 131,072 Gaussians (two 256x256 grids), 25 SH
                         # coefficients per channel (sh_degree 4), batch 16 x 1 target view
                         # (encoder_spfsplatv2.py:240,296-321; config/experiment/spfsplatv2/re10k.yaml:36-37,48)
    "REF10V": (3, 1),    # what the shipped 10-view model trains on: 655,360 Gaussians (ten 256x256 grids), 25 SH
                         # coefficients, 3 scenes x 1 target view per step -- 768 tiles with lists of thousands of entries
                         # (config/experiment/spfsplatv2/re10k_10view.yaml:36-37,48)
}


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


HOST_BINDING = None      # cpulist this process was bound to (main), or None
WIDE_AFFINITY = None     # the affinity it started with (the CPU baseline's threads get it back)


def bind_host(dev) -> None:
    """One process per GPU, its launching thread bound to one L3 group of the CPUs next to it (what `numactl --physcpubind`
    does from outside): the host-bound lines (`--api module`, `--eval-latency`, `--eager`) are two host threads taking turns
    -- the caller's and autograd's, which is created later and inherits the binding --, and the same command ran 0.358 ms
    per step with both in one L3 group and 0.420 with one on each socket: per-process luck otherwise
    (spfsplatv2_amd/hostbind.py).  Called AFTER the synthetic inputs have been made: torch's CPU thread pool exists by then
    and keeps the cores the box has (created under the binding, its 100+ threads would share 16 CPUs).  The graph-replayed
    headline does not care either way.  SPF_BIND=0: leave placement alone."""
    global HOST_BINDING
    if os.environ.get("SPF_BIND", "1") != "0" and HOST_BINDING is None:
        from spfsplatv2_amd.hostbind import bind_to_gpu_l3
        HOST_BINDING = bind_to_gpu_l3(dev.index)
        log(f"launching thread bound to CPUs {HOST_BINDING}" if HOST_BINDING else "host threads not bound (topology not readable)")


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(args, batch_cpu) -> dict:
    """CPU oracle, fwd+bwd, same workload, bounded sample (first scene, one view at a time until ~`--cpu-budget`
    seconds are spent).  The oracle's per-tile tensors are small, so more than ~16 threads only adds
    synchronisation cost: threads = min(host cores, 16); `cores` is what was used, `host_cores` what the box has."""
    from tests import util
    from spfsplatv2_amd import synthetic as syn
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 16)
    if HOST_BINDING and WIDE_AFFINITY:
        os.sched_setaffinity(0, WIDE_AFFINITY)      # (the oracle's thread pool is created now: on the cores the box has)
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
    return {"value": round(done * h * w / spent / 1e6, 5), "unit": "Mpixels/s", "cores": cores,
            "host_cores": host_cores, "kind": "port",
            "sample": f"{done} of the step's {S * V} renders, same workload "
                      f"({batch_cpu.means.shape[1]} Gaussians, {h}x{w}), oracle/splat_ref.py fwd+bwd in float32, "
                      f"{spent:.1f} s on {cores} of the host's {host_cores} cores"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= ranks) of this node; default: WORLD_SIZE if a launcher set it, else 1")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes", type=int, default=None, help="scenes per GPU per step (default: per --config)")
    ap.add_argument("--views", type=int, default=None, help="target views per scene (default: per --config)")
    ap.add_argument("--config", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--s-mult", type=float, default=1.0)
    ap.add_argument("--min-trials", type=int, default=25, help="back-to-back repetitions of the --steps loop")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="GPU work the trials must add up to (3 s: longer than the sampling period of an outside GPU "
                         "activity monitor)")
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
                    help="(default now) capture one whole step (decoder fwd + loss + bwd) in a HIP graph and replay it; "
                         "every --probe-every-th step of the timed region is launched eagerly instead, with HIP events "
                         "around the dominant kernel")
    ap.add_argument("--eager", action="store_true",
                    help="launch every step kernel by kernel from Python (the default until the step's GPU time fell "
                         "below what one Python thread needs to launch it, ~0.45 ms: eager is host-bound now); "
                         "--exact and --allreduce imply it")
    ap.add_argument("--probe-every", type=int, default=10,
                    help="graph mode: every n-th step of the timed region is an eager launch with HIP events around the "
                         "dominant kernel (the roofline's live duration); 0: none (duration from the warm-up survey)")
    ap.add_argument("--streams", type=int, default=1,
                    help="N > 1: the per-GPU batch is split by scene into N micro-batches, each captured in its own HIP "
                         "graph and replayed on its own stream, so that the latency-bound stages of one micro-batch "
                         "(projection, binning, sort) run under the VALU-bound compositing of another (implies --graph; "
                         "same renders, same loss -- each micro-batch weighs 1/N --, same backward)")
    ap.add_argument("--api", default="batched", choices=["batched", "per-view", "module"],
                    help="per-view: the step is the REFERENCE's own glue, unchanged -- `repeat` of every Gaussian tensor "
                         "per view (decoder_splatting_cuda.py:59-64), torch camera preamble, then b*v sequential "
                         "`GaussianRasterizer(settings)(...)` calls with `.item()` host syncs (cuda_splatting.py:96-143) "
                         "-- on this library's drop-in `diff_gauss_pose` surface (eager; what a caller gets with zero "
                         "source changes).  batched (default): one `DecoderSplattingCUDA`-style call for all renders")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default single-GPU C2 run only: skip the `secondary` object (BASELINE configs 3 and 5, the "
                         "reference's real 2-view workload with and without SH band 4, the two-stream C2 step and the "
                         "test_step-shaped decoder-call latency, each measured by a child run of this script)")
    ap.add_argument("--eval-latency", action="store_true",
                    help="measure the LATENCY of one decoder call at the reference's test_step shape instead of the "
                         "training step: b = 1 scene of the 2-view model (131,072 Gaussians, 25 SH coefficients), v = 3 "
                         "target views, 256x256, forward only under no_grad (src/model/model_wrapper.py:415-454)")
    ap.add_argument("--rope", action="store_true",
                    help="measure the RoPE-2D kernel (curope's rope_2d, SURVEY.md 8a row A9) instead of the decoder step: "
                         "fp32 and fp16 at the reference's encoder / decoder attention shapes (48,256,16,64) and "
                         "(32,258,12,64) on strided q views of a qkv buffer, next to the reference's two CPU paths")
    ap.add_argument("--sh-split", action="store_true",
                    help="d_sh = 25 workloads (REF2V / REF10V): the harmonics are resident BAND-SPLIT -- planes [.,3,16] and "
                         "[.,3,9] (Gaussians.harmonics_band4, SpfDims.sh_layout 2), what the fused adapter writes with "
                         "split_harmonics=True -- so that the default degree-3 evaluation never moves band 4's bytes")
    ap.add_argument("--with-adapter", action="store_true",
                    help="the step starts at the network's RAW channels: UnifiedGaussianAdapter.forward (fused HIP pre-pass, "
                         "gaussian_adapter.py:122-150) -> decoder -> MSE -> backward through the adapter to dL/draw: the "
                         "producer of the harmonics layout is inside the timed step (with --sh-split it writes the split)")
    ap.add_argument("--raw-fused", action="store_true",
                    help="like --with-adapter, but the adapter is FUSED INTO THE DECODER (UnifiedGaussianAdapter("
                         "fuse_into_decoder=True), SpfDims.sh_layout 3): the projection kernels apply the adapter's "
                         "activations as they read the raw rows and chain the backward into dL/draw -- no adapter pass")
    ap.add_argument("--allreduce", action="store_true",
                    help="outer-training-step variant (BASELINE config 5): every rank renders its own views of the "
                         "SAME scenes and the Gaussian-parameter gradients are summed with one RCCL all-reduce")
    return ap.parse_args(argv)


def rank_launch_command(n: int, argv: list[str], port: int) -> list[str]:
    """The command `--gpus N` runs when no launcher set WORLD_SIZE: one process per GPU on this node, exactly the
    driver's own form (torch.distributed.run, 127.0.0.1 rendezvous)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def launch_ranks(args, argv: list[str]) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks here.  Never falls back to one GPU."""
    have = torch.cuda.device_count()
    if have < args.gpus and not args.one_device:
        print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s); refusing to report a smaller job "
              "(--one-device --backend gloo exercises the multi-rank path on one GPU)", file=sys.stderr)
        return 2
    if args.one_device and args.backend == "nccl":
        print("bench.py: --one-device needs --backend gloo (RCCL refuses two ranks on one GPU)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = rank_launch_command(args.gpus, argv, port)
    log("no launcher (WORLD_SIZE unset): " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    return subprocess.call(cmd, env=env)


def valu_roofline(kernel: str, launch_ms: float, default_workload, chunks: int = 1):
    """VALU-issue roofline of the dominant kernel: a wave64 VALU instruction occupies its SIMD for 4 cycles, so
    frac = wave-instructions x 4 / (CUs x SIMDs x clock x launch time).  Wave-instructions per launch come from the SQ
    counter pass committed under profiles/ (SQ_INSTS_VALU in sq_summary_<config>.json, collected on exactly that config's
    default workload; `default_workload` = the config's name, or None when the run is not that workload)."""
    prof = ROOT / "profiles" / f"sq_summary_{default_workload}.json"
    if not (default_workload and prof.exists()):
        return None
    try:
        insts = json.loads(prof.read_text())[kernel]["SQ_INSTS_VALU_per_launch"] / chunks   # (counted on whole-call launches)
    except Exception:
        return None
    peak = CU_COUNT * SIMD_PER_CU * CLOCK_HZ / 4.0                 # wave-instructions / s the chip can issue
    ach = insts / (launch_ms * 1e-3)
    return {"bound": "valu-issue", "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 2),
            "unit": "G wave-instructions/s", "frac": round(ach / peak, 5), "wave_instructions_per_launch": insts}


SECONDARY = (
    # name, extra argv, extra environment
    ("C3", ["--config", "C3"], {}),
    ("C5", ["--config", "C5"], {}),
    ("REF2V", ["--config", "REF2V"], {"SPF_SH_BAND4": "0"}),
    ("REF2V_band4", ["--config", "REF2V"], {"SPF_SH_BAND4": "1"}),
    ("REF10V", ["--config", "REF10V"], {"SPF_SH_BAND4": "0"}),
    # the same two shapes with the harmonics BAND-SPLIT (what the fused adapter writes on request): decoder-only steps,
    # comparable with REF2V / REF10V; and with the adapter INSIDE the step, both layouts (the producer's cost in the open)
    ("REF2V_split", ["--config", "REF2V", "--sh-split"], {"SPF_SH_BAND4": "0"}),
    ("REF10V_split", ["--config", "REF10V", "--sh-split"], {"SPF_SH_BAND4": "0"}),
    ("REF2V_adapter", ["--config", "REF2V", "--with-adapter"], {"SPF_SH_BAND4": "0"}),
    ("REF2V_adapter_split", ["--config", "REF2V", "--with-adapter", "--sh-split"], {"SPF_SH_BAND4": "0"}),
    ("REF10V_adapter_split", ["--config", "REF10V", "--with-adapter", "--sh-split"], {"SPF_SH_BAND4": "0"}),
    # ... and with the adapter fused into the projection kernels (sh_layout 3): the same step, no adapter pass at all
    ("REF2V_raw_fused", ["--config", "REF2V", "--raw-fused"], {"SPF_SH_BAND4": "0"}),
    ("REF10V_raw_fused", ["--config", "REF10V", "--raw-fused"], {"SPF_SH_BAND4": "0"}),
    ("C2_stress", ["--config", "C2", "--s-mult", "10"], {}),     # SURVEY.md 8(d)'s stress regime: footprints x 10, dense tiles
    ("C2_module", ["--config", "C2", "--api", "module"], {}),    # an UNCHANGED caller: DecoderSplattingCUDA.forward, nothing configured
    ("C2_streams2", ["--config", "C2", "--streams", "2"], {}),
    ("eval_1x3", ["--eval-latency"], {"SPF_SH_BAND4": "0"}),
    ("rope2d", ["--rope"], {}),
)


def run_secondary(args) -> dict:
    """The other workloads under the same clock as the headline: one child run of this script each (its own process:
    a failure cannot take the headline's NUMBER down, and every child gets the library state of a fresh start), >= 1 s of
    timed GPU work per child, no CPU baseline (rope2d carries its own).  Returns {name: trimmed child line + wall_s}; a
    child that failed leaves {"error": ...} there AND its name in `out["_failed"]`: main() prints the line and then exits
    non-zero, so a broken secondary workload cannot pass a driver that only looks at the return code."""
    out = {}
    env0 = {k: v for k, v in os.environ.items()
            if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                         "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    for name, extra, env_extra in SECONDARY:
        cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--no-cpu-baseline", "--no-secondary", "--min-seconds", "1.0", "--min-trials", "5",
               *extra]
        t0 = time.perf_counter()
        try:
            # (a child starts with the affinity this process STARTED with and binds itself: its CPU baselines included)
            widen = (lambda: os.sched_setaffinity(0, WIDE_AFFINITY)) if WIDE_AFFINITY else None
            r = subprocess.run(cmd, env=dict(env0, **env_extra), capture_output=True, text=True, timeout=180,
                               preexec_fn=widen)
            if r.returncode != 0:
                raise RuntimeError(f"exit code {r.returncode}: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''}")
            line = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:                                  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300], "wall_s": round(time.perf_counter() - t0, 2)}
            out.setdefault("_failed", []).append(name)
            log(f"secondary {name}: FAILED ({out[name]['error']})")
            continue
        keep = {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better") if k in line}
        keep["workload"] = line.get("config", {}).get("workload")
        keep["launch"] = line.get("config", {}).get("launch")
        if "roofline" in line:
            rf = line["roofline"]
            keep["dominant_kernel"] = {k: rf.get(k) for k in ("kernel", "launch_ms", "achieved", "frac", "frac_by_counters",
                                                               "traffic", "algorithmic_bytes_per_launch")}
        for k in ("latency_ms", "timing", "cases", "cpu_baseline", "roofline"):
            if k in line and not (k == "roofline" and name != "rope2d"):
                keep[k] = line[k]
        keep["wall_s"] = round(time.perf_counter() - t0, 2)
        out[name] = keep
        log(f"secondary {name}: {keep.get('value')} {keep.get('unit')} ({keep['wall_s']} s)")
    return out


def eval_latency(args, dev) -> dict:
    """One decoder call at the shape the reference evaluates with (`test_step`: b = 1, v = 3 target views rendered from
    the 2-view model's 131,072 Gaussians with 25 SH coefficients, forward only, src/model/model_wrapper.py:415-454):
    wall time from the call to its result being complete, i.e. with a device synchronisation per call.  The call is the
    reference's own: `decoder.forward(gaussians, extrinsics, intrinsics, near, far, (h, w))` on the decoder MODULE
    (`DecoderSplattingCUDA` under the registry name "splatting_cuda") -- which runs planned calls that will not be
    differentiated on a forward-only PREPARED step (state at fixed addresses, structs built once, inputs bound per call;
    `planned_median`: what a caller gets without doing anything) and, for plans that cannot be prepared, keeps a cache of
    captured HIP graphs (`planned_graph_cache_median`: that cache alone, round 5's path; `planned_no_graph_cache_median`:
    the general launcher, kernel by kernel; `planned_graph_replay_*`: a graph captured by the CALLER around the call)."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import decoder as dec, synthetic as syn
    b = syn.make_batch("REF2V", 1, 3, seed=4242).to(dev)
    bind_host(dev)
    h, w = b.image_shape
    G, K = b.means.shape[1], b.harmonics.shape[-1]
    decoder = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.0, 0.0, 0.0],
                                                          make_scale_invariant=True, enable_cov_grad=False,
                                                          enable_sh_grad=False)).to(dev)
    gaussians = dec.Gaussians(b.means, b.covariances, b.rotations, b.scales, b.harmonics, b.opacities)

    def call():
        with torch.no_grad():
            return decoder.forward(gaussians, b.extrinsics, b.intrinsics, b.near, b.far, (h, w))

    def latency(fn, n):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3, ts[0] * 1e3

    n = max(args.steps * 10, 100)
    unchanged_caller = decoder.auto_plan              # (the module's default: it plans for itself)
    decoder.auto_plan = None                          # exact mode, every call
    for _ in range(max(args.warmup, 3)):
        call()
    exact_med, exact_min = latency(call, n)
    reference_image = call().color.clone()
    plan = spf.plan_pair_budget(decoder.last_call, slack=1.25, check="deferred")
    # what an UNCHANGED caller gets: no max_pairs, the module's own planning + graph cache (first call exact, then planned)
    decoder.auto_plan = unchanged_caller
    for _ in range(max(args.warmup, 3) + 2):
        call()
    default_med, default_min = latency(call, n)
    if not torch.equal(call().color, reference_image):
        raise RuntimeError("the unchanged-caller path does not reproduce the exact-mode image")
    decoder.auto_plan = None
    decoder.clear_eval_graphs()
    decoder.clear_prepared_steps()
    decoder.max_pairs = plan
    decoder.eval_graphs = decoder.prepare_steps = False          # the general launcher
    for _ in range(max(args.warmup, 3)):
        call()
    nograph_med, nograph_min = latency(call, n)
    decoder.eval_graphs = True                                   # the module's graph cache alone (round 5's path)
    for _ in range(max(args.warmup, 3)):
        call()
    if not decoder._graphs:
        raise RuntimeError("the decoder module did not capture the evaluation call")
    cache_med, cache_min = latency(call, n)
    decoder.clear_eval_graphs()
    decoder.eval_graphs = False
    graph_med = graph_min = None
    try:
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            call()
        for _ in range(3):
            g_.replay()
        graph_med, graph_min = latency(g_.replay, n)
        del g_
    except Exception as e:                                      # noqa: BLE001
        log(f"eval latency: graph capture failed ({type(e).__name__}: {e})")
    decoder.eval_graphs = decoder.prepare_steps = True          # the module as it comes
    for _ in range(max(args.warmup, 3)):
        call()
    if not (decoder._prepared_steps or decoder._graphs):
        raise RuntimeError("the decoder module neither prepared nor captured the evaluation call")
    plan_med, plan_min = latency(call, n)
    if not torch.equal(call().color, reference_image):
        raise RuntimeError("the prepared evaluation call does not reproduce the exact-mode image")
    # back-to-back planned calls (no wait between them): what an evaluation loop that only reads the images later sees
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    m = 0
    while m < n or time.perf_counter() - t0 < 1.0:
        call()
        m += 1
    torch.cuda.synchronize(dev)
    stream_ms = (time.perf_counter() - t0) / m * 1e3
    if spf.plan_flags(decoder.last_call) != 0:
        raise RuntimeError("the planned pair budget did not hold in the evaluation-shape run")
    best = plan_med
    return {"metric": "decoder forward latency, b=1 x v=3 at 256x256 (test_step shape)", "value": round(best, 4),
            "unit": "ms", "higher_is_better": False, "n_gpus": 1, "steps": n, "warmup": args.warmup,
            "ms_per_step": round(best, 4), "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"eval_1x3: 1 scene of {G} Gaussians, {K} SH coefficients per channel, 3 target views, "
                                   f"{h}x{w}, DecoderSplattingCUDA.forward under no_grad",
                       "launch": "one decoder.forward call, then torch.cuda.synchronize: wall time per call; planned "
                                 "calls run on the module's forward-only prepared step",
                       "host_binding": HOST_BINDING and f"process bound to CPUs {HOST_BINDING} (one L3 group next to the GPU)"},
            "latency_ms": {"exact_mode_median": round(exact_med, 4), "exact_mode_min": round(exact_min, 4),
                           "unchanged_caller_median": round(default_med, 4), "unchanged_caller_min": round(default_min, 4),
                           "planned_median": round(plan_med, 4), "planned_min": round(plan_min, 4),
                           "planned_graph_cache_median": round(cache_med, 4), "planned_graph_cache_min": round(cache_min, 4),
                           "planned_no_graph_cache_median": round(nograph_med, 4),
                           "planned_no_graph_cache_min": round(nograph_min, 4),
                           "planned_graph_replay_median": None if graph_med is None else round(graph_med, 4),
                           "planned_graph_replay_min": None if graph_min is None else round(graph_min, 4),
                           "planned_back_to_back": round(stream_ms, 4), "calls_each": n,
                           "Mpixels_per_s_back_to_back": round(3 * h * w / stream_ms / 1e3, 1)}}


ROPE_SHAPES = ((48, 256, 16, 64), (32, 258, 12, 64))     # encoder self-attention, decoder


ROPE_COLD_BYTES = 512 << 20     # bytes of q rows a rotation must touch before it comes back to a buffer (2 x the 256 MiB MALL)


def rope_bench(args, dev) -> dict:
    """RoPE-2D (`curope.rope_2d`, curope.cpp:49-65 / kernels.cu:84-108) under the driver's clock: the HIP kernel in place
    on strided q views (and q + k in one launch) of [B,N,3,H,D] qkv buffers, as blocks.py:97-104 calls it.

    The reference rotates a DIFFERENT q / k of a different layer every time (24 encoder + 12 decoder blocks per forward):
    every call reads its rows from HBM.  So `us` = device time per call in a ROTATION over n distinct qkv buffers whose q
    rows add up to > 512 MiB (twice the 256 MiB Infinity Cache: a buffer's lines are cold again when its turn comes):
    one HIP graph of one call per buffer, replayed, HIP events around the replays.  `us_cache_resident` (round 5's method,
    50 dependent calls on ONE buffer: the working set stays in the Infinity Cache) is kept next to it, labelled as what it
    is -- the kernel's issue-side ceiling, not an HBM figure.  Algorithmic bytes: 2*B*N*H*D*sizeof +
    16*B*N per tensor; `roofline.traffic` = the PMC bytes per launch of the headline case (profiles/pmc_summary_rope.json,
    tools/pmc_rope.sh: `--rope --eager` runs only that case, launched one by one).  CPU baselines, fp32, same shapes:
    oracle/rope_ref.c (the restated C++ loop curope.cpp:11-47, one core, as the reference runs it) and
    oracle/rope_torch_ref.py (the restated PyTorch fallback pos_embed.py:112-159, on up to 16 host cores)."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import _lib
    cases = []
    calls = 50
    pmc_only = args.eager            # counter passes: the headline case only, eager launches over the rotation
    for (B, N, H, D) in ROPE_SHAPES[:1] if pmc_only else ROPE_SHAPES:
        for dt in (torch.float32,) if pmc_only else (torch.float32, torch.float16):
            gen = torch.Generator().manual_seed(B + N)
            esz = 4 if dt == torch.float32 else 2
            q_bytes = B * N * H * D * esz
            n_buf = max(8, -(-ROPE_COLD_BYTES // q_bytes))
            base = torch.randn(B, N, 3, H, D, generator=gen).to(dev, dt)
            bufs = [base] + [base.clone() for _ in range(n_buf - 1)]
            pos = torch.randint(0, 18, (B, N, 2), generator=gen).to(dev)
            qs = [t[:, :, 0] for t in bufs]                      # strided [B,N,H,D] views, stride(2) = D
            ks = [t[:, :, 1] for t in bufs]
            for pair in (False,) if pmc_only else (False, True):
                def one(i):
                    if pair:
                        spf.rope_2d_pair(qs[i], ks[i], pos, 100.0, 1.0)
                    else:
                        spf.rope_2d(qs[i], pos, 100.0, 1.0)
                for i in range(n_buf):
                    one(i)
                if pmc_only:
                    for _ in range(3):
                        for i in range(n_buf):
                            one(i)
                    torch.cuda.synchronize(dev)
                    byts = 2 * q_bytes + 16 * B * N
                    return {"metric": "RoPE-2D counter pass (headline case only, eager rotation)", "value": None,
                            "config": {"workload": f"rope2d fp32 q ({B},{N},{H},{D}) over {n_buf} buffers"},
                            "algorithmic_bytes": byts}
                _lib.stage_timing_enable(["rope2d"])
                for i in range(min(20, n_buf)):
                    one(i)
                torch.cuda.synchronize(dev)
                ms, cnt = _lib.stage_times()["rope2d"]
                _lib.stage_timing_enable(False)
                single = 1e3 * ms / max(cnt, 1)
                reps = max(args.steps // 2, 10)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

                def timed(graph, per_replay):
                    graph.replay()
                    torch.cuda.synchronize(dev)
                    ts = []
                    for _ in range(5):
                        e0.record()
                        for _ in range(reps):
                            graph.replay()
                        e1.record()
                        torch.cuda.synchronize(dev)
                        ts.append(e0.elapsed_time(e1) * 1e3 / (reps * per_replay))
                    return sorted(ts)[len(ts) // 2]
                g_rot = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_rot):
                    for i in range(n_buf):
                        one(i)
                us = timed(g_rot, n_buf)
                g_hot = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_hot):
                    for _ in range(calls):
                        one(0)
                us_hot = timed(g_hot, calls)
                del g_rot, g_hot
                byts = (2 if pair else 1) * 2 * q_bytes + 16 * B * N
                cases.append({"shape": [B, N, H, D], "dtype": str(dt).split(".")[-1],
                              "tensors": "q+k, one launch" if pair else "q", "us": round(us, 3),
                              "buffers_in_rotation": n_buf, "q_bytes_in_rotation": n_buf * q_bytes * (2 if pair else 1),
                              "us_cache_resident": round(us_hot, 3), "us_single_event": round(single, 3),
                              "algorithmic_bytes": byts, "GBs": round(byts / us / 1e3, 1),
                              "frac": round(byts / us / 1e3 / HBM_PEAK_GBS, 4),
                              "GBs_cache_resident": round(byts / us_hot / 1e3, 1)})
                log(f"rope2d {cases[-1]}")
            del bufs, qs, ks, base
            torch.cuda.empty_cache()
    # CPU baselines (fp32, q only): the reference's two CPU implementations, restated under oracle/
    from oracle import rope_torch_ref
    from tests import util
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 16)
    if HOST_BINDING and WIDE_AFFINITY:
        os.sched_setaffinity(0, WIDE_AFFINITY)      # (the CPU baselines run on the cores the box has)
    torch.set_num_threads(cores)
    lib = util.rope_oracle_lib()
    cpu = []
    for (B, N, H, D) in ROPE_SHAPES:
        gen = torch.Generator().manual_seed(B + N)
        tok = torch.randn(B, N, H, D, generator=gen)
        pos = torch.randint(0, 18, (B, N, 2), generator=gen)
        byts = 2 * B * N * H * D * 4 + 16 * B * N
        t0 = time.perf_counter()
        n_c = 0
        while n_c < 2 or time.perf_counter() - t0 < 2.0:
            lib.rope2d_ref_f32(tok.data_ptr(), pos.data_ptr(), B, N, H, D, tok.stride(0), tok.stride(1), 100.0, 1.0)
            n_c += 1
        t_c = (time.perf_counter() - t0) / n_c
        bhnd = tok.transpose(1, 2).contiguous()
        cache = {}
        rope_torch_ref.rope2d_fallback(bhnd, pos, 100.0, cache)
        t0 = time.perf_counter()
        n_t = 0
        while n_t < 3 or time.perf_counter() - t0 < 2.0:
            rope_torch_ref.rope2d_fallback(bhnd, pos, 100.0, cache)
            n_t += 1
        t_t = (time.perf_counter() - t0) / n_t
        cpu.append({"shape": [B, N, H, D], "dtype": "float32",
                    "c_loop": {"us": round(t_c * 1e6, 1), "GBs": round(byts / t_c / 1e9, 3), "cores": 1, "calls": n_c,
                               "kind": "port", "what": "oracle/rope_ref.c = curope.cpp:11-47 restated"},
                    "torch_fallback": {"us": round(t_t * 1e6, 1), "GBs": round(byts / t_t / 1e9, 3), "cores": cores,
                                       "calls": n_t, "kind": "port",
                                       "what": "oracle/rope_torch_ref.py = pos_embed.py:112-159 restated (out of place)"}})
        log(f"rope2d cpu {cpu[-1]}")
    head = cases[0]
    worst = min(cases, key=lambda c: c["frac"])
    traffic = None
    prof = ROOT / "profiles" / "pmc_summary_rope.json"
    if prof.exists():
        try:
            traffic = json.loads(prof.read_text())["spf_rope2d_vec_kernel"]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
    return {"metric": "RoPE-2D in place, GB/s of algorithmic bytes (fp32 q at (48,256,16,64))", "value": head["GBs"],
            "unit": "GB/s", "higher_is_better": True, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["us"] / 1e3, 6), "scaling": "weak", "vs_baseline": None, "dtype": "f32 / f16",
            "data": "synthetic",
            "config": {"workload": "rope2d: curope.rope_2d on strided q (and q + k) views of a [B,N,3,H,D] qkv buffer, "
                                   "(B,N,H,D) = (48,256,16,64) and (32,258,12,64), float32 and float16, positions 0..17",
                       "launch": f"one call per buffer of a rotation over {head['buffers_in_rotation']}+ distinct qkv buffers "
                                 "(> 512 MiB of q rows between two visits of a buffer: cold lines, as the reference's 36 "
                                 "attention blocks see them) in one HIP graph, replayed; HIP events around the replays"},
            "roofline": {"bound": "hbm", "kernel": "spf_rope2d_vec_kernel", "achieved": head["GBs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": head["frac"], "traffic": traffic, "launch_ms": round(head["us"] / 1e3, 6),
                         "algorithmic_bytes_per_launch": head["algorithmic_bytes"],
                         "frac_of_copy_ceiling": round(head["GBs"] / HBM_COPY_GBS, 4),
                         "cache_resident": {"GBs": head["GBs_cache_resident"], "us": head["us_cache_resident"],
                                            "note": f"{calls} dependent calls on ONE buffer (round 5's method): the working "
                                                    "set sits in the 256 MiB Infinity Cache -- not an HBM figure"},
                         "worst_case": {k: worst[k] for k in ("shape", "dtype", "tensors", "us", "frac")}},
            "cases": cases,
            "cpu_baseline": {"value": cpu[0]["c_loop"]["GBs"], "unit": "GB/s", "cores": 1, "host_cores": host_cores,
                             "kind": "port", "sample": "whole tensors, >= 2 s per implementation and shape", "shapes": cpu}}


def main():
    args = parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a launcher decides the job size; a line that claims another N than was run would be a lie
        if rank == 0:
            print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        sys.exit(2)
    builder = rank == 0 if args.one_device else local_rank == 0
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # a launcher (WORLD_SIZE set) gets a process group even for a world of one: `torch.distributed.run
    # --nproc-per-node=1` is the pre-flight of the RCCL path on a single-GPU box (init with device_id, barrier,
    # flat-bucket all-reduce on device tensors)
    launched = "WORLD_SIZE" in os.environ
    if launched:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # the in-tree library normally travels with the tree; (re)build it if it is missing or stale (one rank per
    # node compiles, the others wait) -- a no-op when the sources' digest matches
    # (SPF_LIB_DIR names a development variant built by hand, e.g. for tools/ab.sh: never rebuilt behind one's back)
    from spfsplatv2_amd import build as _build
    if builder and "SPF_LIB_DIR" not in os.environ:
        _build.build(verbose=False)
    if launched:
        dist.barrier()
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import _lib, synthetic as syn

    global WIDE_AFFINITY
    WIDE_AFFINITY = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None

    if args.eval_latency or args.rope:
        if world != 1:
            sys.exit("bench.py: --eval-latency / --rope are single-GPU measurements")
        print(json.dumps(rope_bench(args, dev) if args.rope else eval_latency(args, dev)), flush=True)
        if launched:
            dist.destroy_process_group()
        return

    S = args.scenes if args.scenes is not None else WORKLOADS[args.config][0]
    V = args.views if args.views is not None else WORKLOADS[args.config][1]
    from spfsplatv2_amd import shard
    if args.allreduce:      # same scenes everywhere, rank-specific target poses
        batch_cpu = syn.make_batch(args.config, S, V, seed=1000, s_mult=args.s_mult)
        pg = torch.Generator().manual_seed(7000 + rank)
        batch_cpu.extrinsics = torch.stack([syn.target_poses(pg, V) for _ in range(S)])
    else:                   # independent scenes per rank (scene-first sharding of a larger batch)
        batch_cpu = syn.make_batch(args.config, S, V, seed=1000 + rank, s_mult=args.s_mult)
    b = batch_cpu.to(dev)
    bind_host(dev)
    h, w = b.image_shape
    G, K = b.means.shape[1], b.harmonics.shape[-1]
    names = ("means", "scales", "rotations", "opacities", "harmonics", "extrinsics")
    bg = torch.zeros(3, device=dev)
    if args.raw_fused:
        args.with_adapter = True
        if args.sh_split:
            sys.exit("bench.py: --raw-fused reads the raw rows themselves: there is no harmonics tensor to lay out")
    if (args.sh_split or args.with_adapter) and (K != 25 or args.allreduce or args.api == "per-view"):
        sys.exit("bench.py: --sh-split / --with-adapter are for the d_sh = 25 workloads (REF2V, REF10V), batched or module api")
    adapter_mod = raw_all = None
    if args.with_adapter:
        # raw network channels that the adapter maps back onto the resident batch (same scales -> same pairs -> the same
        # decoder workload as the plain config): softplus^-1 of the scales, the unit quaternions, harmonics / sh_mask
        from spfsplatv2_amd import adapter as _ad
        adapter_mod = _ad.UnifiedGaussianAdapter(_ad.GaussianAdapterCfg(0.5, 15.0, 4), split_harmonics=args.sh_split,
                                                 fuse_into_decoder=args.raw_fused).to(dev)
        y = (b.scales.double() / 0.001).clamp_min(1e-12)
        raw_scales = torch.where(y > 20.0, y, torch.log(torch.expm1(y.clamp_max(20.0)))).float()
        raw_all = torch.cat((raw_scales, b.rotations,
                             (b.harmonics / adapter_mod.sh_mask.to(dev)).reshape(*b.harmonics.shape[:2], 75)), dim=-1)
        names = ("means", "opacities", "extrinsics", "raw")
    if args.streams > 1:
        if S % args.streams or args.allreduce:
            sys.exit("bench.py: --streams N needs a scene count divisible by N (and is not combined with --allreduce)")
        args.graph = True
    if args.api == "per-view":
        if args.streams > 1 or args.allreduce:
            sys.exit("bench.py: --api per-view is a single-stream, single-rank-style step")
        args.eager, args.exact, args.graph = True, True, False      # host syncs inside the step: nothing to capture
    if args.api == "module":
        # the step of an UNCHANGED caller of the batched decoder module: DecoderSplattingCUDA.forward with nothing
        # configured -- the module plans for itself and verifies every call (one host sync per forward, at its end)
        if args.streams > 1 or args.allreduce:
            sys.exit("bench.py: --api module is a single-stream, single-rank-style step")
        args.eager, args.graph = True, False
    if not args.graph:          # launch mode: HIP-graph replay unless something in the step cannot be captured
        args.graph = not (args.eager or args.exact or args.allreduce)

    class MicroBatch:
        """Scenes [s0, s1) of the resident batch: own leaves, own call record / pair budget, own loss share."""

        def __init__(self, s0, s1):
            self.sl = slice(s0, s1)
            self.leaves = {n: (raw_all if n == "raw" else getattr(b, n))[self.sl].clone().requires_grad_(True) for n in names}
            if args.sh_split and not args.with_adapter:
                full = self.leaves["harmonics"].detach()
                self.leaves["harmonics"] = full[..., :16].contiguous().requires_grad_(True)
                self.leaves["harmonics_band4"] = full[..., 16:].contiguous().requires_grad_(True)
            self.weight = (s1 - s0) / S
            self.record = spf.CallRecord()
            self.max_pairs = None
            # --allreduce: the five Gaussian gradients are written straight into ONE flat bucket (no cat, no copy back)
            # and summed across the ranks by one collective per micro-batch, issued asynchronously: RCCL runs it on
            # its own stream under the next micro-batch's kernels
            self.bucket = shard.GradBucket(*(self.leaves[n] for n in names[:5])) if args.allreduce else None
            self.work = None
            self.decoder = None

        def render_per_view(self):
            """The reference's decoder + render_cuda, statement for statement, on the drop-in rasterizer surface."""
            from math import isqrt
            from spfsplatv2_amd import decoder as dec
            L, sl = self.leaves, self.sl
            bb, vv = L["extrinsics"].shape[:2]
            rep = lambda t: t[:, None].expand(bb, vv, *t.shape[1:]).reshape(bb * vv, *t.shape[1:])   # `repeat`
            ext = L["extrinsics"].reshape(bb * vv, 4, 4)
            near, far = b.near[sl].reshape(-1), b.far[sl].reshape(-1)
            means, scales_, rot = rep(L["means"]), rep(L["scales"]), rep(L["rotations"])
            harm, opac = rep(L["harmonics"]), rep(L["opacities"])
            scale = 1 / near                                                    # cuda_splatting.py:66-74
            ext = ext.clone()
            ext[..., :3, 3] = ext[..., :3, 3] * scale[:, None]
            means = means * scale[:, None, None]
            scales_ = scales_ * scale[:, None, None]
            near, far = near * scale, far * scale
            degree = isqrt(harm.shape[-1]) - 1
            shs = harm.transpose(-1, -2).contiguous()                           # "b g xyz n -> b g n xyz"
            fov_x, fov_y = dec.get_fov(b.intrinsics[sl].reshape(bb * vv, 3, 3)).unbind(dim=-1)
            tan_x, tan_y = (0.5 * fov_x).tan(), (0.5 * fov_y).tan()
            proj = dec.get_projection_matrix(near, far, fov_x, fov_y).transpose(-1, -2)
            view = ext.inverse().transpose(-1, -2)
            images = []
            for i in range(bb * vv):                                            # cuda_splatting.py:96-143
                mean_gradients = torch.zeros_like(means[i], requires_grad=True)
                settings = spf.GaussianRasterizationSettings(
                    image_height=h, image_width=w, tanfovx=tan_x[i].item(), tanfovy=tan_y[i].item(), bg=bg,
                    scale_modifier=1.0, projmatrix=proj[i], sh_degree=degree, prefiltered=False, debug=False,
                    enable_cov_grad=True, enable_sh_grad=True)
                image, _depth, _norm, _alpha, _radii, _extra = spf.GaussianRasterizer(settings)(
                    means3D=means[i], means2D=mean_gradients, shs=shs[i], colors_precomp=None,
                    opacities=opac[i, ..., None], scales=scales_[i], rotations=rot[i], viewmatrix=view[i])
                images.append(image)
            return torch.stack(images).reshape(bb, vv, 3, h, w)

        def step(self):
            L, sl = self.leaves, self.sl
            for t in L.values():
                t.grad = None
            if args.api == "per-view":
                color = self.render_per_view()
                if self.record.get("num_pairs") is None:        # (statistics for the byte model: one batched call)
                    with torch.no_grad():
                        spf.render_views(L["extrinsics"], b.intrinsics[sl], b.near[sl], b.far[sl], (h, w), bg,
                                         L["means"], L["harmonics"], L["opacities"], L["rotations"], L["scales"],
                                         record=self.record)
                loss = spf.mse_loss(color, b.target[sl], self.weight)
                loss.backward(gradient=spf.unit_grad(dev))
                return loss
            if args.with_adapter:
                ga = adapter_mod(L["means"], L["opacities"], L["raw"], with_covariances=False)
                L = dict(L, scales=ga.scales, rotations=ga.rotations, harmonics=ga.harmonics,
                         harmonics_band4=ga.harmonics_band4, fused=ga.raw)
            if args.api == "module":
                from spfsplatv2_amd import decoder as dec
                if self.decoder is None:
                    self.decoder = dec.get_decoder(dec.DecoderSplattingCUDACfg(
                        name="splatting_cuda", background_color=[0.0, 0.0, 0.0], make_scale_invariant=True,
                        enable_cov_grad=True, enable_sh_grad=True)).to(dev)
                    if args.exact:
                        self.decoder.auto_plan = None
                out = self.decoder.forward(dec.Gaussians(L["means"], None, L["rotations"], L["scales"], L["harmonics"],
                                                         L["opacities"], harmonics_band4=L.get("harmonics_band4"),
                                                         raw=L.get("fused")),
                                           L["extrinsics"], b.intrinsics[sl], b.near[sl], b.far[sl], (h, w))
                color = out.color
                if "num_pairs" not in self.record:                  # (the first call of a shape is exact: statistics)
                    self.record.update({k: v for k, v in self.decoder.last_call.items() if k != "counters"})
            else:
                color, depth, _alpha = spf.render_views(
                    L["extrinsics"], b.intrinsics[sl], b.near[sl], b.far[sl], (h, w), bg, L["means"], L["harmonics"],
                    L["opacities"], L["rotations"], L["scales"], scale_invariant=True, enable_cov_grad=True,
                    enable_sh_grad=True, max_pairs=self.max_pairs, record=self.record,
                    gaussian_sh_band4=L.get("harmonics_band4"), gaussian_raw=L.get("fused"))
            if args.torch_loss:
                loss = torch.nn.functional.mse_loss(color, b.target[sl]) * self.weight
            else:
                loss = spf.mse_loss(color, b.target[sl], self.weight)
            # (the cached dL/dloss = 1 saves autograd's fill kernel, and the fused loss recognises it: its backward is
            #  the forward's unit gradient, no launch)
            if args.allreduce:
                with self.bucket:
                    loss.backward(gradient=spf.unit_grad(dev))
                self.work = self.bucket.all_reduce(async_op=True, skip_single=False)
            else:
                loss.backward(gradient=spf.unit_grad(dev))
            return loss

    per = S // args.streams
    if args.allreduce and S > 1:
        # one micro-batch per scene on the one stream: scene i's all-reduce overlaps scene i+1's forward and backward
        micro = [MicroBatch(i, i + 1) for i in range(S)]
    else:
        micro = [MicroBatch(i * per, (i + 1) * per) for i in range(args.streams)]

    def step():
        for m in micro:
            m.step()
        for m in micro:
            if m.work is not None:
                m.work.wait()           # (stream-level wait: the current stream waits for the collective, the host does not)
                m.work = None

    def barrier():
        if launched:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(values: list[float]) -> list[float]:
        if world == 1:
            return values
        t = torch.tensor(values, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    log(f"batch resident: {S} scenes x {V} views, G={G}, K={K}, {h}x{w}")
    step()
    torch.cuda.synchronize(dev)
    if args.api == "module" and not args.exact:
        # the module replays its training calls from its own HIP graphs (a replay launches nothing from the host, so the
        # library's stage events see nothing): the per-stage survey is taken from eager launches first
        _lib.stage_timing_enable(True)
        for m in micro:
            m.decoder.prepare_steps = False
        n_survey = max(args.warmup, 3)
        for _ in range(n_survey):
            step()
        torch.cuda.synchronize(dev)
        module_survey = {k: (v[0] / n_survey, 1) for k, v in _lib.stage_times().items() if k != "rope2d" and v[1] > 0}
        _lib.stage_timing_enable(False)
        for m in micro:
            m.decoder.prepare_steps = os.environ.get("SPF_PREPARE_STEPS", "1") != "0"
    else:
        module_survey = None
    D_total = sum(m.record["num_pairs"] for m in micro)
    log(f"first step done: D={D_total}, max tile list={max(m.record['max_tile_list'] for m in micro)}")
    if (not args.exact or args.graph) and args.api == "batched":
        for m in micro:
            m.max_pairs = spf.plan_pair_budget(m.record, slack=1.25, check="deferred")
        log(f"planned budget: {micro[0].max_pairs}" + (f" x {len(micro)} micro-batches" if len(micro) > 1 else ""))
    max_pairs = micro[0].max_pairs
    run = step
    eager_survey = None
    if args.graph:
        # per-stage survey and dominant-kernel timing need eager launches (events are recorded at launch time,
        # a replayed graph launches nothing from the host); with --streams the surveyed launches are one
        # micro-batch's, timed exclusively (in the timed region the kernels of the streams overlap)
        _lib.stage_timing_enable(True)
        n_survey = max(args.warmup, 3)
        for _ in range(n_survey):
            micro[0].step()
        torch.cuda.synchronize(dev)
        # per STEP (a stage is one launch per chunk of the call, see spf_raster_chunks)
        eager_survey = {k: (v[0] / n_survey, 1) for k, v in _lib.stage_times().items() if k != "rope2d" and v[1] > 0}
        _lib.stage_timing_enable(False)
        streams = [torch.cuda.Stream(dev) for _ in micro] if len(micro) > 1 else [torch.cuda.current_stream(dev)]
        graphs = []
        try:
            for m, st_ in zip(micro, streams):
                for t in m.leaves.values():
                    t.grad = None
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, **({"stream": st_} if len(micro) > 1 else {})):
                    m.step()
                graphs.append(g_)
        except Exception as e:                      # (never seen; a runtime that cannot capture still gets a number)
            if len(micro) > 1:
                raise
            log(f"HIP graph capture failed ({type(e).__name__}: {e}); falling back to eager launches")
            torch.cuda.synchronize(dev)
            args.graph, graphs, eager_survey = False, [], None
        if not graphs:
            pass
        elif len(micro) == 1:
            run = graphs[0].replay
        else:
            def run():
                for g_, st_ in zip(graphs, streams):
                    with torch.cuda.stream(st_):
                        g_.replay()
        if graphs:
            log(f"step captured in {len(graphs)} HIP graph(s)")
    # warm-up doubles as the per-stage survey (HIP events around every stage); the timed region then keeps
    # events only around the dominant kernel, so the headline number is not diluted by 14 event records/step
    _lib.stage_timing_enable(True)
    for _ in range(max(args.warmup, 1)):
        run()
    torch.cuda.synchronize(dev)
    survey = {k: v for k, v in _lib.stage_times().items() if k != "rope2d" and v[1] > 0}
    if eager_survey is None and module_survey is not None:
        eager_survey = module_survey
    if eager_survey is not None:
        survey = {k: (v[0] * max(args.warmup, 1), max(args.warmup, 1)) for k, v in eager_survey.items()}
    dom = max(survey, key=lambda k: survey[k][0])
    _lib.stage_timing_enable(False)

    # graph mode: every probe-th step is the same step launched eagerly (same kernels, same buffers' shapes, same
    # stream) so that the library's HIP events around the dominant kernel exist inside the timed region; the host
    # spends ~0.5 ms on it while the GPU still has the replays before it in its queue
    probe = args.probe_every if (args.graph and len(micro) == 1 and args.probe_every > 0) else 0

    def trial() -> float:
        """EXACTLY --steps steps between barrier + synchronize on both sides."""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if probe and i % probe == probe - 1:
                step()
            else:
                run()
        barrier()
        return time.perf_counter() - t0

    # number of trials: >= --min-trials and >= --min-seconds of GPU work in total, agreed between the ranks from a
    # first (untimed, discarded) trial
    est = max_over_ranks([trial()])[0]
    n_trials = int(min(2000, max(args.min_trials, math.ceil(args.min_seconds / max(est, 1e-6)))))
    # events around every n-th launch of the dominant kernel only (an event pair leaves the GPU idle for ~11 us):
    # at most ~1000 samples over the whole timed region (the library keeps 1024 per stage)
    _lib.stage_timing_enable([dom])
    n_event_launches = n_trials * (args.steps // probe if probe else args.steps)
    _lib.stage_timing_sample_every(max(1 if probe else 4, math.ceil(n_event_launches / 1000)))
    trials = [trial() for _ in range(n_trials)]
    stages = _lib.stage_times()
    _lib.stage_timing_enable(False)
    _lib.stage_timing_sample_every(1)
    if max_pairs is not None and any(spf.plan_flags(m.record) != 0 for m in micro):
        raise RuntimeError(f"the planned pair budget did not hold (flags "
                           f"{[spf.plan_flags(m.record) for m in micro]}): results invalid")
    trials = max_over_ranks(trials)                       # per trial: the slowest rank
    dt = sorted(trials)[len(trials) // 2]                 # median trial
    log(f"timed region: {n_trials} trials x {args.steps} steps = {sum(trials):.3f} s; median trial {dt * 1e3:.3f} ms, "
        f"min {min(trials) * 1e3:.3f}, max {max(trials) * 1e3:.3f}")

    failed_children = []
    if rank == 0:
        P = h * w
        renders = world * S * V
        value = renders * P * args.steps / dt / 1e6
        # the library runs a call as `chunks` launch chains of whole scenes on two streams (spf_raster_chunks): the
        # dominant kernel is launched once per chunk, each launch processes 1/chunks of the step's units
        lib_ = _lib.load()
        chunk_stages = {"bin_pairs": 0, "tile_sort": 0, "render_fwd": 0, "render_bwd": 1, "project_bwd": 1}
        chunks = lib_.spf_raster_chunks(S // len(micro), V, h, w, chunk_stages[dom]) if dom in chunk_stages else 1
        dom_ms = (stages[dom][0] / stages[dom][1]) if stages[dom][1] else survey[dom][0] / survey[dom][1] / chunks
        # (with --streams the surveyed launch is one micro-batch's: its share of the scenes and pairs)
        # band-split harmonics evaluated to degree 3: the algorithm needs (and the kernels move) 16 of the 25 coefficients
        K_model = 16 if (args.sh_split and os.environ.get("SPF_SH_BAND4", "0") != "1") else K
        dom_bytes = stage_bytes(dom, S // len(micro), V, G, K_model, P, D_total // len(micro)) / chunks
        if args.api == "per-view":          # one launch per render
            dom_bytes = stage_bytes(dom, 1, 1, G, K_model, P, D_total // (S * V))
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        # the counter passes under profiles/ (tools/refresh_profiles.sh) were collected per config on its default batch
        default_workload = args.config if ((S, V) == WORKLOADS[args.config] and args.s_mult == 1.0 and
                                           not args.allreduce and len(micro) == 1 and args.api == "batched" and
                                           not args.sh_split and not args.with_adapter) else None
        prof = ROOT / "profiles" / f"pmc_summary_{args.config}.json"
        kernel = _lib.stage_kernel_name(dom)
        if prof.exists() and default_workload:      # the PMC passes were collected on exactly this workload
            try:
                traffic = json.loads(prof.read_text()).get(kernel, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        A = total_bytes(S, V, G, K_model, P, D_total)
        if args.with_adapter and not args.raw_fused:   # + the adapter, both directions: raw in / parameters out, gradients in / dL/draw out
            A += 2.0 * S * G * (2 * (7 + 3 * K) * 4 - (0 if K_model == K else 4 * 3 * (K - K_model)))
        deg = int(K ** 0.5) - 1
        out = {
            "metric": "Mpixels/s fwd+bwd, 256x256 @ ~65k Gaussians" if args.config == "C2" else
                      f"Mpixels/s fwd+bwd ({args.config})",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {G} pixel-aligned Gaussians/scene, {K} SH coefficient(s) per "
                                   f"channel (sh_degree {deg}), {h}x{w}, {S} scenes x {V} views per GPU per step, "
                                   + ("Gaussian adapter FUSED INTO the projection kernels (raw network channels in) + " if args.raw_fused
                                      else "fused Gaussian adapter (raw network channels in) + " if args.with_adapter else "")
                                   + "decoder fwd + MSE + bwd to all Gaussian parameters and poses"
                                   + (" and through the adapter to the raw channels" if args.with_adapter else "")
                                   + ("; harmonics band-split [.,3,16] | [.,3,9] (sh_layout 2)" if args.sh_split else ""),
                       "loss": "torch.nn.functional.mse_loss" if args.torch_loss else "spfsplatv2_amd.mse_loss (fused HIP)",
                       "host_binding": HOST_BINDING and f"process bound to CPUs {HOST_BINDING} (one L3 group next to the GPU)",
                       "scenes_per_gpu": S, "views_per_scene": V, "gaussians_per_scene": G, "image": [h, w],
                       "sh_coeffs": K, "renders_per_step": renders, "pairs_per_render": round(D_total / (S * V), 1),
                       "s_mult": args.s_mult, "pair_buffer": ("module default: planned by DecoderSplattingCUDA itself, verified per call (one host sync at the "
                                       "end of each forward)" if args.api == "module" and not args.exact else
                                       "exact (read-back per step)") if max_pairs is None else
                                      f"planned from step 0 (x1.25 = {max_pairs.capacity} pairs), verified on device",
                       "launch": (f"{len(micro)} micro-batches of {S // len(micro)} scenes, one HIP graph and one stream "
                                  "each (kernels of the streams overlap; roofline durations are exclusive, from an "
                                  "eager survey of one micro-batch)" if len(micro) > 1 else
                                  (f"hip-graph replay of the whole step; every {probe}th step launched eagerly with HIP "
                                   "events around the dominant kernel" if probe else "hip-graph replay")
                                  if args.graph else "eager"),
                       "process_group": (dist.get_backend() if launched else None),
                       "api": ("per-view: the reference's own glue (repeat + torch camera preamble + b*v sequential "
                               "GaussianRasterizer calls with .item() syncs) on the drop-in surface"
                               if args.api == "per-view" else
                               ("module: DecoderSplattingCUDA.forward with nothing configured by the caller"
                                if args.api == "module" else "batched: one decoder call for all renders")),
                       "sharding": ("views of the same scenes per rank + RCCL all-reduce of the Gaussian gradients: one flat "
                                    "bucket per scene micro-batch, written in place by the backward kernels, reduced "
                                    "asynchronously under the next micro-batch"
                                    if args.allreduce else "scene-first, no data-path collective")},
            "timing": {"trials": n_trials, "statistic": "median trial; each trial = exactly `steps` steps between "
                                                        "barrier+synchronize, max over ranks",
                       "seconds_timed": round(sum(trials), 4)},
            "trials_ms": [round(t * 1e3, 4) for t in trials],
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": None if traffic is None else traffic / chunks, "launch_ms": round(dom_ms, 5),
                         # the same fraction from the COUNTER bytes (FETCH_SIZE doubled + WRITE_SIZE per launch): where the
                         # kernel moves fewer bytes than SURVEY.md 8(d)'s model charges (REF2V / REF10V: the projection
                         # backward), this is the smaller, honest figure of what crosses HBM
                         "frac_by_counters": None if traffic is None else
                         round(traffic / chunks / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "launches_per_step": chunks,
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "valu": valu_roofline(kernel, dom_ms, default_workload, chunks),
                         "path_achieved_GBs": round(A / (dt / args.steps) / 1e9, 2),
                         "path_frac_of_copy_ceiling": round(A / (dt / args.steps) / 1e9 / HBM_COPY_GBS, 5)},
            "stage_ms_per_step_warmup": {k: round(v[0] / max(args.warmup, 1), 5) for k, v in survey.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, batch_cpu)
        headline_defaults = (args.config == "C2" and default_workload == "C2" and not args.eager and not args.exact
                             and not args.torch_loss)
        if world == 1 and headline_defaults and not args.no_secondary:
            torch.cuda.empty_cache()
            out["secondary"] = run_secondary(args)
            failed_children = out["secondary"].pop("_failed", [])
        print(json.dumps(out), flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()
    if failed_children:
        # the headline line is out (its number is valid on its own); a broken secondary workload still fails the run
        sys.exit(f"bench.py: secondary workload(s) failed: {', '.join(failed_children)}")


if __name__ == "__main__":
    main()
