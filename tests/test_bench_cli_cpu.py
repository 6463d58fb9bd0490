"""bench.py's launcher half, which runs before any GPU call and therefore here: `--gpus N` without a rank environment
must start N ranks, and must REFUSE (exit code != 0, a message that says why) when fewer than N devices are visible
instead of quietly measuring one GPU and printing n_gpus: 1."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FRCNN_BENCH_BACKEND")}
    env.update(extra_env)
    return subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_gpus_flag_refuses_without_enough_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices visible: the refusal path needs fewer")
    r = _run({})
    assert r.returncode == 2 and "refusing" in r.stderr and "--gpus 2" in r.stderr, (r.returncode, r.stderr[-500:])
    assert "{" not in r.stdout                       # no JSON line that could be mistaken for a measurement


def test_gpus_flag_gloo_still_needs_a_device():
    import torch
    if torch.cuda.device_count() >= 1:
        import pytest
        pytest.skip("a device is visible")
    r = _run({"FRCNN_BENCH_BACKEND": "gloo"})
    assert r.returncode == 2 and "no HIP device" in r.stderr


def test_default_pass_shape():
    """What `python bench.py` (the driver's command) and its documented variants replay: images per captured pass x passes in flight."""
    sys.path.insert(0, ROOT)
    import bench
    f = bench.default_pass_shape
    assert f("c2", False) == (4, 4)                           # the headline: four fp32 images per pass, four passes
    assert f("c2", False, streams=1) == (1, 1)                # --streams 1: the latency form, one image
    assert f("c2", False, streams=12) == (1, 12)              # naming --streams keeps one image per pass
    assert f("c2", False, batch=1) == (1, 12)                 # --batch 1: twelve one-image passes on twelve queues
    assert f("c2", False, batch=8, streams=3) == (8, 3)
    assert f("c2", False, no_graph=True) == (1, 12)           # eager runs: one image per pass unless --batch says otherwise
    assert f("c2", False, batch=4, streams=1, no_graph=True) == (4, 1)     # (the PMC passes of scripts/profile_round5.sh)
    assert f("c4", True) == (8, 4) and f("c2", True) == (8, 4)             # bf16: eight per pass, four passes
    assert f("c4", True, streams=4) == (8, 4)
    assert f("c1", False) == (4, 4)                           # configs[0]: RPN only; four images per pass since round 6 (1 x 12: 710 img/s, 4 x 4: 767)
    assert f("c1", False, batch=1) == (1, 12)


def test_extra_legs_are_compacted_and_survive_a_failed_child(monkeypatch):
    """`python bench.py` (N = 1, configs[1]) carries the other BASELINE configs under `extra`: child processes, started before the parent
    touches the GPU.  Here (no GPU) every child fails: each leg must come back as {"error": ..., "seconds": ...}, never an exception; the
    compaction of a child's line keeps value / config / roofline and drops the rest."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "EXTRA_LEGS", (("c0_vgg16_rpn", ["bench.py", "--config", "c1", "--no-extra", "--steps", "1", "--warmup", "0"]),
                                             ("train_steps_f32", ["-c", "import sys; sys.exit(3)"])))
    got = bench.other_config_legs(timeout_s=120)
    assert set(got) == {"c0_vgg16_rpn", "train_steps_f32"}
    for leg in got.values():
        assert "error" in leg and "seconds" in leg
    line = {"metric": "m", "value": 1.0, "unit": "img/s", "ms_per_step": 2.0, "steps": 3, "warmup": 1, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "w", "graphs_in_flight": 4, "n_rois_kept": 300},
            "roofline": {"bound": "mfma", "achieved": 10.0, "peak": 2500.0, "frac": 0.004, "unit": "TFLOP/s", "kernel": "k",
                         "all_conv_launches": {"achieved": 5.0, "frac": 0.002, "launches": [1, 2, 3]}, "end_to_end_conv_tflops": 7.0},
            "with_host_io": {"value": 0.5}}
    c = bench.compact_inference_leg(line)
    assert c["value"] == 1.0 and c["config"]["workload"] == "w" and "n_rois_kept" not in c["config"] and "with_host_io" not in c
    assert c["roofline"]["frac"] == 0.004 and c["roofline"]["all_conv_launches"] == {"achieved": 5.0, "frac": 0.002}
    t = bench.compact_train_leg({"dtype": "f32", "workload": "w", "losses_read": "late", "step_launch": "g",
                                 "rpn_step1": {"ms_per_step": 2.0, "img_s": 500.0, "roofline": {"frac": 0.5}, "grad_payload_MB": 47.3,
                                               "through_loop": {"fast_feed": {"ms_per_iteration": 2.1, "iterations": 64, "distinct_images": 32}}}})
    assert t["rpn_step1"]["through_train_util_loop"]["ms_per_iteration"] == 2.1 and "det_step2" not in t and "grad_payload_MB" not in t["rpn_step1"]
