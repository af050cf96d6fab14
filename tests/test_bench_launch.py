"""CPU: `python bench.py --gpus N` must be startable the way the driver starts it -- without a launcher around it (the
N > 1 line could not even be started).  The launcher re-executes itself under torch.distributed.run on 127.0.0.1; here the ranks are
gloo processes on CPU (`--cpu-standin`: the launch / rendezvous / report plumbing without kernels, labelled `standin: true`)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600, env=e, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


@pytest.mark.parametrize("mode,bf16", [("allreduce", False), ("rs_ag", False), ("rs_ag", True)])
def test_gpus_2_self_launches_two_ranks_and_prints_one_json_line_last(mode, bf16):
    rc, lines, err = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-standin", "--comm-mode", mode, *(["--comm-bf16"] if bf16 else []))
    assert rc == 0, err[-2000:]
    out = json.loads(lines[-1])                      # the JSON line is the LAST line of stdout
    assert out["n_gpus"] == 2 and out["standin"] is True and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 * 64 and out["config"]["parallelism"] == "dp2"
    c = out["comm"]
    assert c["ranks"] == 2 and c["backend"] == "gloo" and c["mode"] == mode and c["bf16"] is bf16
    assert c["replicas_identical"] is True and c["bytes_per_step"] > 0 and c["collectives_per_step"] >= 2
    assert sum(1 for ln in lines if ln.startswith("{") and '"metric"' in ln) == 1


@pytest.mark.parametrize("mode,bf16", [("allreduce", False), ("rs_ag", True)])
def test_gpus_8_self_launches_the_node_the_headline_names(mode, bf16):
    """BASELINE's metric is quoted at 1 / 2 / 4 / 8 GPUs: the same self-launch at the real rank count (8 gloo ranks on CPU) -- rendezvous,
    per-rank seeds, the bucket partition at world 8 (rs_ag needs bucket lengths divisible by 8), one JSON line, replicas identical"""
    rc, lines, err = _run("--gpus", "8", "--steps", "2", "--warmup", "1", "--cpu-standin", "--comm-mode", mode, *(["--comm-bf16"] if bf16 else []),
                          env={"OMP_NUM_THREADS": "1"})
    assert rc == 0, err[-2000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 8 and out["standin"] is True and out["config"]["global_batch"] == 8 * 64 and out["config"]["parallelism"] == "dp8"
    c = out["comm"]
    assert c["ranks"] == 8 and c["mode"] == mode and c["bf16"] is bf16 and c["replicas_identical"] is True
    assert c["collectives_per_step"] >= (4 if mode == "rs_ag" else 2)
    assert sum(1 for ln in lines if ln.startswith("{") and '"metric"' in ln) == 1


def test_too_few_devices_gives_a_json_error_line_not_a_traceback():
    if __import__("torch").cuda.is_available() and __import__("torch").cuda.device_count() >= 64:
        pytest.skip("a 64-GPU box")
    rc, lines, err = _run("--gpus", "64", "--steps", "1", "--warmup", "0")
    assert rc != 0
    out = json.loads(lines[-1])
    assert out["value"] is None and out["n_gpus"] == 64 and "GPU" in out["error"]


def test_world_size_mismatch_is_reported_as_json():
    rc, lines, err = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-standin",
                          env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert rc != 0
    out = json.loads(lines[-1])
    assert out["value"] is None and "WORLD_SIZE" in out["error"]


def test_traffic_is_quoted_only_for_matching_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from the committed PMC passes -- and only while the hash of mtp_amd/csrc recorded in that file equals the tree's
    (VERDICT r02 #7): another hash -> None plus the reason in traffic_source"""
    import json
    import bench
    from tools.pmc_hbm import csrc_sha
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench._traffic_from_profiles("gemm_nt") == (None, None)                       # no file at all
    (prof / "r07_pmc_hbm.json").write_text(json.dumps({"_commit": "abc", "_csrc_sha": "0" * 16, "gemm_nt_kernel": {"hbm_bytes_per_launch": 123}}))
    val, why = bench._traffic_from_profiles("gemm_nt")
    assert val is None and "other kernel sources" in why
    (prof / "r08_pmc_hbm.json").write_text(json.dumps({"_commit": "def", "_csrc_sha": csrc_sha(), "gemm_nt_kernel": {"hbm_bytes_per_launch": 456}}))
    val, why = bench._traffic_from_profiles("gemm_nt")                                   # the newest file, matching hash
    assert val == 456 and "r08_pmc_hbm.json" in why and "def" in why


def test_clock_sampler_without_a_device_reports_nothing(monkeypatch):
    import glob
    import bench
    monkeypatch.setattr(glob, "glob", lambda pattern: [])
    s = bench.ClockSampler(0)
    s.start()
    assert s.stop() is None


def test_grouped_weight_gradient_planning():
    """tiles and pieces of the grouped weight-gradient launches (host side of mtp_gemm_tn_grouped)"""
    from mtp_amd import ops
    assert ops.grouped_tiles(1024, 4096) == 64 and ops.grouped_tiles(192, 192) == 1 and ops.grouped_tiles(264, 520) == 6
    assert ops.grouped_tiles(108, 192) == 0 and ops.grouped_tiles(4, 256) == 0                 # not multiples of 8: the split-K kernel
    assert ops.grouped_splits(12544, 48) == 1 and ops.grouped_splits(8192, 9) == 1             # a transformer block's / InternImage level 2's problems: whole contraction
    assert ops.grouped_splits(50176, 64) == 4                                                    # ViT FPN, second deconvolution: 4 x 64 tiles = one round
    assert ops.grouped_splits(131072, 1) == 32 and ops.grouped_splits(32768, 4) == 8           # InternImage levels 0 / 1
    assert ops.grouped_splits(524288, 1) == 64                                                   # the stem: capped


def test_rccl_log_parser_and_report_order():
    """`comm.rccl` is a tolerant scan of rank 0's NCCL_DEBUG=INFO log (what RCCL chose: channels, algorithm, protocol, transport); and the passes behind
    the `comm` object (timed collectives, the same steps without them) and behind `roofline` (single-stream instrumented steps) run AFTER the timed region
    whose time `value` is computed from -- they cannot change it"""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    with tempfile.NamedTemporaryFile("w", suffix=".log", delete=False) as f:
        f.write("h:1:1 [0] NCCL INFO RCCL version 2.22.3+hip7.0\n"
                "h:1:2 [0] NCCL INFO Channel 00/16 : 0 1 2 3 4 5 6 7\nh:1:2 [0] NCCL INFO Channel 15/16 : 0 1 2 3 4 5 6 7\n"
                "h:1:2 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC\n"
                "h:1:2 [0] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels\n"
                "h:1:2 [0] NCCL INFO AllReduce: 209715200 Bytes -> Algo 1 proto 2 time 0.1 Ring Simple\n")
    info = b.parse_rccl_log(f.name)
    os.unlink(f.name)
    assert info["version"].startswith("2.22") and info["channels"] == 16 and info["algorithms"] == ["Ring"] and info["protocols"] == ["Simple"]
    assert "P2P/IPC" in info["transports"] and info["collective_lines"] == 1
    # the forms seen on the MI355X box (round 5): the banner's "RCCL version : x", and NCCL's TUNING line with the algorithm / protocol as numbers only
    with tempfile.NamedTemporaryFile("w", suffix=".log", delete=False) as f:
        f.write("RCCL version : 2.26.6-HEAD:64f48b6\nh:1:2 [0] NCCL INFO Channel 127/128 : 0\n"
                "h:1:2 [0] NCCL INFO ReduceScatter: 26214400 Bytes -> Algo 1 proto 0 time 55.0\nh:1:2 [0] NCCL INFO AllReduce: 1024 Bytes -> Algo 0 proto 1 time 9.0\n")
    info = b.parse_rccl_log(f.name)
    os.unlink(f.name)
    assert info["version"].startswith("2.26.6") and info["channels"] == 128 and info["algorithms"] == ["Ring", "Tree"] and info["protocols"] == ["LL", "LL128"]
    assert len(info["tuning_lines"]) == 2
    assert b.parse_rccl_log("/nonexistent/file") is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    i_value = main.index("value = world * B * args.steps / dt")
    assert i_value < main.index("red.timing, red.timed = True, []") and i_value < main.index("forward_only = None")
    assert main.index("dt = time.perf_counter() - t0") < main.index("eng_cls.wgrad_side_stream = False")     # the attribute is only touched after the timed region ...
    assert "finally:\n            eng_cls.wgrad_side_stream = side_default" in main                             # ... and always restored
