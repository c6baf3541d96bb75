"""bench.py's launcher logic (no GPU needed): `--gpus N` never degrades to a smaller job."""
import os
import subprocess
import sys

from tests.conftest import ROOT


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rank_launch_command_is_the_drivers_form():
    b = _bench()
    cmd = b.rank_launch_command(4, ["--gpus", "4", "--steps", "7"], 12345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")


def test_gpus_n_without_enough_devices_fails_loudly():
    """No silent single-GPU number: here there is no GPU at all, so --gpus 2 must exit non-zero and print
    no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        return
    assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_workload_table_covers_baseline_configs():
    b = _bench()
    from spfsplatv2_amd import synthetic as syn
    assert set(b.WORKLOADS) == {"C2", "C3", "C5", "REF2V", "REF10V"} and all(k in syn.CONFIGS for k in b.WORKLOADS)
    assert syn.CONFIGS["REF2V"][0] == 2 * 256 * 256 and syn.CONFIGS["REF2V"][4] == 25
    # byte model: SURVEY 8(d) total = sum of the stages
    S, V, G, K, P, D = 2, 3, 1000, 4, 4096, 5000
    stages = ("project_fwd", "bin_pairs", "tile_sort", "render_fwd", "render_bwd", "project_bwd")
    assert sum(b.stage_bytes(s, S, V, G, K, P, D) for s in stages) > 0
    assert b.total_bytes(S, V, G, K, P, D) == S * V * (G * (308 + 36 * K) + 60.0 * P) + 124.0 * D


def test_secondary_children_parse_and_cover_the_verdict_list():
    """The default single-GPU run reports every BASELINE config, the reference's real workload (with and without SH
    band 4), the two-stream step and the test_step-shaped latency under `secondary`; every child's command line must be
    accepted by the script's own parser and must not recurse."""
    b = _bench()
    names = [n for n, _, _ in b.SECONDARY]
    assert names == ["C3", "C5", "REF2V", "REF2V_band4", "REF10V", "REF2V_split", "REF10V_split", "REF2V_adapter",
                     "REF2V_adapter_split", "REF10V_adapter_split", "REF2V_raw_fused", "REF10V_raw_fused", "C2_stress", "C2_module", "C2_streams2", "eval_1x3",
                     "rope2d"]
    for _name, extra, env in b.SECONDARY:
        a = b.parse_args(["--gpus", "1", "--no-cpu-baseline", "--no-secondary", *extra])
        assert a.no_secondary and a.no_cpu_baseline
        assert all(isinstance(k, str) and isinstance(v, str) for k, v in env.items())
    assert dict((n, e) for n, _, e in b.SECONDARY)["REF2V_band4"] == {"SPF_SH_BAND4": "1"}


def test_a_failed_secondary_child_is_recorded_and_named():
    """VERDICT r4 (robustness 13): a secondary child that dies must not hide in stdout -- run_secondary names it in
    `_failed`, and main() exits non-zero after printing the headline line.  The child here dies in its argument parser
    (no GPU needed)."""
    b = _bench()
    b.SECONDARY = (("broken", ["--no-such-flag"], {}),)
    out = b.run_secondary(b.parse_args(["--steps", "2", "--warmup", "1"]))
    assert out["_failed"] == ["broken"] and "error" in out["broken"] and "exit code" in out["broken"]["error"]
