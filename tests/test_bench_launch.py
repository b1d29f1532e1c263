"""bench.py --gpus N without torch.distributed.run (SURVEY 8e / BASELINE config 5).

CPU tier: the launcher itself (rank environment, rendezvous on 127.0.0.1, rank 0 owns stdout, a dying rank takes the job down) with a
tiny gloo script standing where bench.py stands.  GPU tier: the real `python bench.py --gpus 2` on the 1-GPU box, two ranks sharing the
device over gloo -- the whole N > 1 code path (barriers, max over ranks, the 49 MB gradient all-reduce inside sds_step), one JSON line."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


RANK_SCRIPT = textwrap.dedent('''
    import os, sys, json
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if "--die" in sys.argv and rank == 1:
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(1) * (rank + 1)
    dist.all_reduce(t)
    print("noise from rank", rank)                      # only rank 0's stdout may reach the launcher's stdout
    dist.barrier()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "sum": float(t.item()), "launcher": os.environ.get("AC_BENCH_LAUNCHER"),
                          "addr": os.environ["MASTER_ADDR"], "local_rank": os.environ["LOCAL_RANK"]}), flush=True)
    dist.destroy_process_group()
''')


def _run_launcher(tmp_path, extra):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    driver = tmp_path / "drive.py"
    driver.write_text(textwrap.dedent(f'''
        import sys, importlib.util
        spec = importlib.util.spec_from_file_location("bench_module", {os.path.join(ROOT, "bench.py")!r})
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        m.torch.cuda.is_available = lambda: True
        m.torch.cuda.device_count = lambda: 3
        sys.exit(m.self_launch(3, {extra!r}, script={str(script)!r}))
    '''))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["AC_BENCH_GRACE_S"] = "3"
    return subprocess.run([sys.executable, str(driver)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)


def test_self_launch_spawns_ranks_and_keeps_stdout_to_rank0(tmp_path):
    r = _run_launcher(tmp_path, [])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    rec = json.loads(lines[-1])                                    # the JSON line is the last thing on stdout
    assert rec == {"n_gpus": 3, "sum": 6.0, "launcher": "self", "addr": "127.0.0.1", "local_rank": "0"}
    assert not any("rank 1" in l or "rank 2" in l for l in lines)  # the other ranks' stdout went to stderr
    assert "noise from rank 1" in r.stderr and "noise from rank 2" in r.stderr


def test_self_launch_propagates_a_dead_rank(tmp_path):
    r = _run_launcher(tmp_path, ["--die"])
    assert r.returncode != 0 and "rank 1 exited with 7" in r.stderr


def test_gpus_without_world_size_goes_to_the_launcher(monkeypatch):
    """`--gpus 8` with no WORLD_SIZE in the environment must reach self_launch, not exit (bench.py:324 of round 3)"""
    m = _load_bench()
    seen = {}
    monkeypatch.setattr(m, "self_launch", lambda n, argv, script=None: seen.update(n=n, argv=list(argv)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        m.main()
    assert e.value.code == 0 and seen == {"n": 8, "argv": ["--gpus", "8", "--steps", "3"]}


@pytest.mark.gpu
def test_bench_gpus_2_self_launched_over_gloo_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["AC_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeat", "1", "--sds-steps", "1",
                        "--posed-frames", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dist_backend"] == "gloo" and rec["launcher"] == "self"
    assert rec["steps"] == 2 and rec["value"] > 0 and rec["scaling"] == "weak"
    lo, hi = rec["ms_per_step_rank_min_max"]
    assert 0 < lo <= hi
    sds = rec["sds_step"]
    assert "error" not in sds, sds
    assert sds["grad_allreduce_mb"] == pytest.approx(48.99, abs=0.02) and sds["grad_allreduce_ms"] > 0
    assert "posed_frame" not in rec and "cpu_baseline" not in rec
    # round 5: the N > 1 line says how the ranks were started, what one rank does alone in the same job, and the collective's bus bandwidth
    assert rec["hsa_ipc_mode_legacy"]["value"] == "0" and "attempt 1" in rec["hsa_ipc_mode_legacy"]["set_by"]
    solo = rec["same_job_solo"]
    assert solo["rays_per_s"] > 0 and 0 < solo["value_over_n_times_solo"] < 1.5 and solo["sds_ms_per_step"] > 0
    assert sds["allreduce_busbw_gbs"] > 0 and sds["allreduce_busbw_peak_gbs"] == pytest.approx(7 * 153.0) and sds["solo_over_n_rank_step_time"] > 0


def _fracs(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == "frac" and isinstance(v, (int, float)):
                yield path, v
            else:
                yield from _fracs(v, f"{path}.{k}")
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _fracs(v, f"{path}[{i}]")


@pytest.mark.gpu
def test_bench_line_names_its_binding_resources_and_no_frac_exceeds_one():
    """the driver's line carries roofline.issue (VALU + fp32-MFMA issue busy fraction) and roofline.gather (per-CU texture-address
    path) from the committed counter pass, the regular-grid kernels' fraction is taken against the 64-byte sectors actually requested of L2, and no `frac`
    anywhere in the record exceeds 1"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--repeat", "1", "--sds-steps", "1", "--posed-frames", "0",
                        "--no-cpu-baseline", "--no-occupancy", "--no-viewdirs", "--no-fine-view", "--sd-arch-steps", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    roof = rec["roofline"]
    assert roof["issue"]["bound"] == "valu+mfma issue" and 0 < roof["issue"]["busy_frac"] and roof["issue"]["valu_insts_per_launch"] > 1e8
    assert roof["gather"]["bound"].startswith("L1 gather") and 0 < roof["gather"]["busy_frac"] <= 1
    over = [(p, v) for p, v in _fracs(rec) if v > 1.0]
    assert not over, over
    m = rec["mesh_export_512"]["roofline"]
    assert m["frac"] <= 1.0 and m["request_rate_vs_hbm_peak"] > 0 and "issue" in m
