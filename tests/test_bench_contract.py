"""bench.py is the driver's measuring stick: guard its command line and the JSON line it prints."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_command_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_bench_refuses_to_run_without_a_gpu():
    """No silent CPU fallback: without a GPU the bench must fail, and loudly."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode != 0
    assert not any(line.startswith("{") for line in out.stdout.splitlines())


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus N` without a launcher re-executes under torch.distributed.run with every flag forwarded; the
    command is built by one function, checked here; --gpus 1 stays a plain in-process run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("k5_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--workload", "10s_nabla", "--no-vae"]
    cmd = b.self_launch_command(argv, 4, port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                                  # every flag forwarded, in order
    free = b.self_launch_command(argv, 2)
    assert 1024 <= int(free[free.index("--master-port") + 1]) < 65536
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"WORLD_SIZE" not in os.environ and args.gpus > 1' in src      # one GPU never takes the launcher path


def test_bench_oversubscribe_flags_select_the_ipc_transport():
    """`bench.py --gpus P --oversubscribe` = P ranks on the devices that exist over the engine's IPC transport (RCCL refuses ranks that share a device):
    the flag sets K5_OVERSUBSCRIBE / K5_SP_TRANSPORT for the ranks it launches, is forwarded to them, refuses RCCL, and its line says INVALID_AS_BENCH."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'os.environ["K5_OVERSUBSCRIBE"] = "1"' in src and 'os.environ["K5_SP_TRANSPORT"] = args.transport' in src
    assert src.index('os.environ["K5_OVERSUBSCRIBE"] = "1"') < src.index("cmd = self_launch_command(sys.argv[1:], args.gpus)")   # before the ranks are spawned
    assert "--oversubscribe needs the IPC transport" in src and '"ipc_ranks_seen"' in src and "ranks time-slice" in src
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--transport", "rccl"], capture_output=True,
                         text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "--oversubscribe needs the IPC transport" in out.stderr


def test_bench_self_launch_reports_missing_devices_after_spawning():
    """On a box with fewer devices than --gpus the ranks are really spawned and each says what is missing (no usage message, rc != 0)."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices are present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-vae",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode != 0
    assert "torch.distributed.run" in out.stderr and "launching" in out.stderr
    assert "needs 2 devices on this node, found" in out.stderr
    assert not any(line.startswith("{") for line in out.stdout.splitlines())


@pytest.mark.gpu
def test_bench_json_line_contract():
    """One JSON line with the contract's keys; two visual blocks keep it short (the engine path is the same)."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X (torch.cuda.is_available() is False)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--blocks", "2",
                          "--no-vae"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "steps/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] / c["value"] > 10.0          # sanity only: the ratio says nothing about kernel quality


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_over_the_ipc_transport():
    """`python bench.py --gpus 2 --oversubscribe`: the bench launches itself under torch.distributed.run, both ranks sit on the one device, the
    engine's IPC group carries K / V^T — and the line says so: n_gpus 2, ipc_ranks_seen 2, no timed-out flag wait, identical latents on both ranks
    (rank_check), rccl_ranks_seen -1 (a communicator that is not RCCL's), and INVALID_AS_BENCH (two ranks time-slice one GPU)."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X (torch.cuda.is_available() is False)")
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one device: --oversubscribe would spread the ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "K5_SP_TRANSPORT", "K5_OVERSUBSCRIBE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "1", "--warmup", "1", "--blocks", "2",
                          "--no-vae", "--no-breakdown"], capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ipc_ranks_seen"] == 2 and d["rccl_ranks_seen"] == -1
    assert d["ipc_transport"]["flag_wait_timeouts"] == 0 and d["ipc_transport"]["ranks_per_device"] == 2 and d["ipc_transport"]["collectives"] > 0
    assert d["rank_check"]["latent_checksums_identical_on_all_ranks"] is True
    assert any("time-slice" in x for x in d["INVALID_AS_BENCH"]) and "sequence-parallel x2" in d["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_measures_its_own_hbm_traffic():
    """`roofline.traffic` of the default line is measured by the run itself (two rocprofv3 --pmc child passes, FETCH_SIZE x2 + WRITE_SIZE, one counter
    per pass): the figure must be a plausible per-launch byte count for the 47 616-token dense attention — at least the algorithmic K / V^T / Q / O
    bytes (0.68 GB), far below re-reading K and V^T for every 256-query workgroup from HBM (62 GB) — and the source string must say it is this run's."""
    import importlib.util
    import shutil
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X (torch.cuda.is_available() is False)")
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("rocprofv3 is not installed on this box")
    spec = importlib.util.spec_from_file_location("k5_bench_lt", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    traffic, src = b.live_traffic(47616)
    if traffic is None:      # the profiler is an external tool: its absence or refusal on a box is not the engine's failure (bench.py then quotes the committed figure)
        pytest.skip(f"rocprofv3 --pmc pass unavailable here: {src}")
    assert 0.6e9 < traffic < 8e9, (traffic, src)
    assert src.startswith("THIS run")
