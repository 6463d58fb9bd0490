"""Multi-GPU readiness on a 1-GPU box: the two programs the driver launches under torch.distributed.run are started
here exactly that way (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment, one process per rank), with two
ranks sharing cuda:0 over the gloo transport (FRCNN_BENCH_BACKEND=gloo; on a node the same code paths run one rank per
GPU over RCCL).  Checks the contract of the JSON line (whole-job value, n_gpus, weak scaling, max-over-ranks timing
reached through the barrier + all-reduce) and that the data-parallel training step reports its one collective."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, world, timeout=900):
    """Start `world` ranks of a script like torch.distributed.run does; return rank 0's last stdout line as JSON."""
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FRCNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world == 1:
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k)
        procs.append(subprocess.Popen([sys.executable] + script_args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    lines = [l for l in outs[0][1].strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][1][-2000:]            # rank 0 prints ONE JSON line
    assert not [l for l in outs[-1][1].strip().splitlines() if l.startswith("{")] or world == 1      # other ranks print none
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu():
    flags = ["bench.py", "--steps", "3", "--warmup", "1", "--streams", "2", "--no-cpu-baseline"]
    one = _launch(flags + ["--gpus", "1"], 1)
    two = _launch(flags + ["--gpus", "2"], 2)
    for line, n in ((one, 1), (two, 2)):
        assert line["n_gpus"] == n and line["scaling"] == "weak" and line["higher_is_better"] is True
        assert line["unit"] == "img/s" and line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
        assert line["steps"] == 3 and line["warmup"] == 1
        assert "replicas x%d" % n in line["config"]["parallelism"] and "configs[1]" in line["config"]["workload"]
        # value is the WHOLE job: ranks x images in flight x steps / (max-over-ranks) time
        assert abs(line["value"] - n * 2 * 3 / (line["ms_per_step"] * 3e-3)) < 0.02 * line["value"]
        assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1
        assert line["with_host_io"]["value"] > 0
        bb = line["roofline"]["backbone_conv"]
        # (three conv_block pairs -- branch2a + shortcut -- share a launch: 43 layers, 40 launches)
        assert bb["layers_per_image"] == 43 and bb["launches_per_image"] == 40 and 0 < bb["frac"] < 1 and abs(bb["gflop_per_image"] - 75.25) < 0.1
        # the backbone with the images in flight is measured on one-rank runs only (rank 0 alone would skew a 2-rank job)
        assert ("in_flight" in bb) == (n == 1) and (n != 1 or (bb["in_flight"]["images_in_flight"] == 2 and 0 < bb["in_flight"]["frac"] < 1))
    # two ranks share ONE GPU here, so the aggregate stays in the neighbourhood of the single-rank figure
    assert 0.5 < two["value"] / one["value"] < 1.6, (one["value"], two["value"])


def test_bench_train_two_ranks_on_one_gpu():
    flags = ["scripts/bench_train.py", "--steps", "2", "--warmup", "1", "--only", "rpn"]
    one = _launch(flags, 1)
    two = _launch(flags, 2)
    assert one["world"] == 1 and one["rpn_step1"]["allreduce_ms"] == 0.0
    assert two["world"] == 2 and two["backend"] == "gloo"
    r = two["rpn_step1"]
    assert abs(r["grad_payload_MB"] - 47.3) < 0.2                    # 11.83 M trainable parameters, fp32
    # the step's ONE collective, timed stand-alone (gloo through host memory with two ranks on one GPU and two timed
    # steps: the stand-alone figure is noisy and may exceed the step it is compared with; on RCCL it is a fraction)
    assert r["allreduce_ms"] > 0 and 0 < r["allreduce_share"] < 10
    assert r["roofline"]["gflop_per_step"] > 200 and 0 < r["roofline"]["frac"] < 1
    assert abs(r["img_s"] - 2 * 1e3 / r["ms_per_step"]) < 0.02 * r["img_s"]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with NO rank environment (how a plain driver command line reads): bench.py starts the
    two ranks itself -- here both on the one GPU over gloo -- relays ONE line with n_gpus 2, and that line carries the
    data-parallel training steps whose gradient went through the collective on both ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FRCNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--streams", "2", "--no-cpu-baseline", "--no-io"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "replicas x2" in line["config"]["parallelism"]
    assert abs(line["value"] - 2 * 2 * 3 / (line["ms_per_step"] * 3e-3)) < 0.02 * line["value"]
    # the self-audit of an N > 1 line: every rank seen, each rank's own rate, the value from the slowest
    rk = line["ranks"]
    assert rk["ranks_seen"] == 2 and len(rk["per_rank_img_s"]) == 2 and rk["slowest_over_fastest_seconds"] >= 1.0
    assert abs(line["value"] - 2 * min(rk["per_rank_img_s"])) < 0.02 * line["value"]
    t = line["train_dp"]
    assert "error" not in t, t
    assert t["ranks_seen"] == 2 and t["backend"] == "gloo" and t["collectives_per_step"] == 1
    assert abs(t["grad_payload_MB"] - 47.3) < 0.2
    assert t["allreduce_ms"] > 0 and t["exposed_allreduce_ms"] >= 0 and t["ms_per_step"] >= t["ms_per_step_without_allreduce"] * 0.9
    assert abs(t["img_s"] - 2 * 1e3 / t["ms_per_step"]) < 0.02 * t["img_s"]
    _check_det_leg(t["det_step2"], ranks=2, backend="gloo")
    # --gpus 1 stays a plain single-process run: no train_dp object
    one = _launch(["bench.py", "--steps", "2", "--warmup", "1", "--streams", "2", "--no-cpu-baseline", "--no-io", "--gpus", "1"], 1)
    assert one["n_gpus"] == 1 and "train_dp" not in one


def _check_det_leg(d, ranks, backend):
    """BASELINE configs[4] inside `train_dp`: detector step-2 steps in mixed bf16, the 89.0 MB f32 gradient payload
    (22.25 M trainable parameters) through the same ONE collective per step."""
    assert "configs[4]" in d["workload"] and "mixed bf16" in d["workload"]
    assert d["ranks_seen"] == ranks and d["backend"] == backend and d["collectives_per_step"] == 1
    assert abs(d["grad_payload_MB"] - 89.0) < 0.2
    assert d["allreduce_ms"] > 0 and d["ms_per_step"] > 0 and d["exposed_allreduce_ms"] >= 0
    assert abs(d["img_s"] - ranks * 1e3 / d["ms_per_step"]) < 0.02 * d["img_s"]


def test_bench_gpus_flag_config_c4_prints_the_same_contract():
    """`python bench.py --gpus 2 --config c4`: configs[3]'s inference replicas (bf16, batched graphs) under the same launcher,
    the same line keys, and the same two training legs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FRCNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "c4", "--steps", "2", "--warmup", "1", "--streams", "1", "--batch", "2",
                        "--no-cpu-baseline", "--no-io"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dtype"] == "bf16" and "configs[3]" in line["config"]["workload"] and line["scaling"] == "weak"
    assert line["config"]["images_per_graph"] == 2 and "replicas x2" in line["config"]["parallelism"]
    assert abs(line["value"] - 2 * 2 * 2 / (line["ms_per_step"] * 2e-3)) < 0.02 * line["value"]            # ranks x images per step x steps / time
    t = line["train_dp"]
    assert "error" not in t and abs(t["grad_payload_MB"] - 47.3) < 0.2 and "configs[2]" in t["workload"]
    _check_det_leg(t["det_step2"], ranks=2, backend="gloo")


def test_bench_multi_rank_path_over_real_rccl():
    """bench.py's multi-rank code path -- process group brought up after the hipGraph captures, barrier-bracketed timing with the
    max over ranks taken on a device tensor, the `train_dp` leg with its asynchronous all-reduce -- over the REAL "nccl" (RCCL)
    backend, with one rank on this box's one GPU (FRCNN_BENCH_FORCE_DIST=1): what a node run does per rank, minus the peers."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FRCNN_BENCH_BACKEND")}
    env.update(FRCNN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--streams", "2", "--no-cpu-baseline", "--no-io"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = r.stdout.strip().splitlines()
    assert len(out_lines) == 1 and out_lines[0].startswith("{"), r.stdout[-1500:]     # RCCL's version banner (written to fd 1) must not reach stdout
    line = json.loads(out_lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0
    t = line["train_dp"]
    assert "error" not in t, t
    assert t["backend"] == "nccl" and t["ranks_seen"] == 1 and abs(t["grad_payload_MB"] - 47.3) < 0.2
    assert t["allreduce_ms"] > 0 and t["ms_per_step"] > 0 and t["exposed_allreduce_ms"] >= 0
    _check_det_leg(t["det_step2"], ranks=1, backend="nccl")


def test_one_rank_through_the_multi_rank_path_measures_what_the_plain_run_measures():
    """the N = 1 value taken THROUGH the multi-rank code path (process group over RCCL, barriers, gathered per-rank
    clocks; FRCNN_BENCH_FORCE_DIST=1) is the plain N = 1 value to within 2 % -- the distributed bracket adds nothing to the timed
    region -- and the line audits itself: one rank seen, its own rate equal to the value."""
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FRCNN_BENCH_BACKEND", "FRCNN_BENCH_FORCE_DIST")}
    flags = [sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-io", "--no-extra", "--no-train-dp"]

    def run(force):
        env = dict(base, FRCNN_BENCH_NO_NATIVE="1", FRCNN_BENCH_NO_ENTRY="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if force:
            env.update(FRCNN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        r = subprocess.run(flags, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])

    plain, forced = run(False), run(True)
    rk = forced["ranks"]
    assert "ranks" not in plain and rk["ranks_seen"] == 1 and abs(rk["per_rank_img_s"][0] - forced["value"]) < 0.005 * forced["value"]
    best_plain, best_forced = plain["value"], forced["value"]
    if abs(best_forced - best_plain) > 0.02 * best_plain:         # one more pair before calling it a difference (box noise is ~1 %)
        best_plain, best_forced = max(best_plain, run(False)["value"]), max(best_forced, run(True)["value"])
    assert abs(best_forced - best_plain) <= 0.02 * best_plain, (best_plain, best_forced)
