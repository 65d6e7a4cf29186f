"""`python bench.py --gpus N` with no launcher around it starts its own N ranks (child processes, before anything touches
the GPU) and returns their exit code -- the form in which the driver's scaling run calls it.  Here on CPU: the launcher,
the rendezvous, the backend agreement, the shard plan of the library's host planner and one halo exchange (`--plumbing-only`:
everything of the N > 1 path except the kernels); on the GPU box (`-m gpu`): the real cfg4 strong-scaling line from two
ranks that share the box's one GPU (gloo dry run), with the reassembled shard outputs compared with rank 0's single-GPU
result of the same run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr.decode()[-2000:]


def test_bench_starts_its_own_ranks_and_exchanges_halos():
    rc, line, err = _run(["--gpus", "2", "--plumbing-only", "--n-times", "6000"])
    assert rc == 0, err
    assert line["plumbing_only"] and line["n_gpus"] == 2 and line["backend"] == "gloo"
    assert line["halo_rows_exact"]
    (a0, a1), (b0, b1) = line["need"]
    assert a1 > 3000 and b0 < 3000  # both ranks fetched rows of the other


def test_cfg5_six_field_halo_exchange_between_two_ranks():
    """cfg5 on several GPUs is a strong-scaling run too (BASELINE.json configs[4]): the six fields' rows travel along axis 1 of the
    [6, rows, modes] storage.  Plumbing only here; the line with its in-run parity figure is the `-m gpu` test below."""
    rc, line, err = _run(["--gpus", "2", "--plumbing-only", "--workload", "cfg5", "--n-times", "1200"])
    assert rc == 0, err
    assert line["plumbing_only"] and line["workload"] == "cfg5" and line["halo_rows_exact"] and line["n_times_total"] == 1200
    (a0, a1), (b0, b1) = line["need"]
    assert a1 > 600 and b0 < 600


def test_failed_rccl_bring_up_is_agreed_on_by_all_ranks_and_falls_back_to_gloo():
    """RCCL asked for where it cannot come up (no GPUs here): every rank publishes its outcome in the ranks' own key-value store,
    reads everybody's, and all of them switch to gloo together -- no rank is left waiting in a collective."""
    rc, line, err = _run(["--gpus", "2", "--plumbing-only", "--n-times", "6000"], SCRI_AMD_BENCH_BACKEND="nccl")
    assert rc == 0, err
    assert line["backend"] == "gloo" and "nccl bring-up failed on rank 0" in line["backend_note"] and line["halo_rows_exact"]


def test_launcher_hands_back_a_failing_rank_exit_code():
    rc, line, err = _run(["--gpus", "2", "--plumbing-only", "--n-times", "3"])  # a 3-sample series has no cubic spline: the ranks raise
    assert rc != 0 and line is None


def test_committed_pmc_summary_is_reported_only_for_the_build_it_was_taken_on(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    stamp = bench.csrc_hash()
    assert len(stamp) == 16 and stamp == bench.csrc_hash()
    good = tmp_path / "good.json"
    good.write_text(json.dumps({"_meta": {"csrc_hash": stamp}, "bms::zgemm3m_mfma_kernel<false>": {"traffic_bytes": 123.0}}))
    stale = tmp_path / "stale.json"
    stale.write_text(json.dumps({"_meta": {"csrc_hash": "0" * 16}, "bms::zgemm3m_mfma_kernel<false>": {"traffic_bytes": 123.0}}))
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(good))
    assert bench.committed_pmc_traffic("cfg3", 1, 100_000)[0] == 123.0
    assert bench.committed_pmc_traffic("cfg3", 2, 100_000)[0] is None
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(stale))
    value, note = bench.committed_pmc_traffic("cfg3", 1, 100_000)
    assert value is None and "not reported" in note


@pytest.mark.gpu
def test_two_rank_cfg4_line_carries_its_own_parity_check():
    rc, line, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--n-times", "60000"], timeout=900)
    assert rc == 0, err
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["ranks"]["world_size"] == 2
    par = line["strong_scaling"]["parity"]
    assert par["rows_compared"] == par["rows_of_n1_result"] > 59000
    assert par["within_bar"], par
    assert line["strong_scaling"]["sharded_vs_n1_max_abs_diff"] <= 1e-14 * par["scale_max_abs"]


@pytest.mark.gpu
def test_two_rank_cfg5_strong_scaling_line_carries_its_own_parity_check():
    """`--workload cfg5 --gpus N`: total fixed (here 4000 steps instead of 2e5), rank 0 transforms the whole six-field series
    first, the two shards' outputs are gathered and compared with it (scri/asymptotic_bondi_data/transformations.py:391-412 is what
    the shards reproduce); two ranks share the box's one GPU (gloo dry run)."""
    rc, line, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg5", "--n-times", "4000"], timeout=900)
    assert rc == 0, err
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["ranks"]["world_size"] == 2
    ss = line["strong_scaling"]
    assert ss["n_times_total"] == 4000 and ss["n1_same_run"]["ms_per_step"] > 0
    par = ss["parity"]
    assert par["rows_compared"] == par["rows_of_n1_result"] > 3900
    assert par["within_bar"], par
