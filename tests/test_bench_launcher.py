"""`python bench.py --gpus N` without a launcher starts the N ranks itself.  Tested here through
`--dry-launch`, in which every child only joins a gloo group and reports its rank: no GPU, no HIP library.  What is under
test is the parent: it must not import torch, must hand every child RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, must relay
rank 0's JSON line, and must exit non-zero when a child dies or when --gpus and WORLD_SIZE disagree."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + argv, env=e, capture_output=True, text=True, timeout=240)


def _json_lines(text):
    out = []
    for line in text.splitlines():
        try:
            d = json.loads(line)
            if isinstance(d, dict):
                out.append(d)
        except ValueError:
            pass
    return out


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(['--gpus', '2', '--dry-launch'])
    assert r.returncode == 0, r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1                                    # ONE line, rank 0's
    assert lines[0]['dry_launch'] is True and lines[0]['n_gpus'] == 2 and lines[0]['ranks_seen'] == [0, 1]
    assert lines[0]['local_rank'] == 0 and lines[0]['master_port']


def test_three_ranks_and_one_rank():
    r = _run(['--gpus', '3', '--dry-launch'])
    assert r.returncode == 0, r.stderr
    assert _json_lines(r.stdout)[0]['ranks_seen'] == [0, 1, 2]
    r = _run(['--dry-launch'])
    assert r.returncode == 0, r.stderr
    assert _json_lines(r.stdout)[0]['n_gpus'] == 1


def test_a_child_that_dies_fails_the_job_and_the_other_ranks_are_stopped():
    r = _run(['--gpus', '2', '--dry-launch'], ASR_BENCH_DRY_FAIL_RANK='1')
    assert r.returncode == 3                                  # the dead rank's own code
    assert 'rank 1 exited with code 3' in r.stderr
    assert not _json_lines(r.stdout)                          # rank 0 never got past the rendezvous: no result line
    r = _run(['--gpus', '2', '--dry-launch'], ASR_BENCH_DRY_FAIL_RANK='0')
    assert r.returncode == 3


def test_world_size_and_gpus_must_agree_under_an_external_launcher():
    r = _run(['--gpus', '1', '--dry-launch'], RANK='0', WORLD_SIZE='2', LOCAL_RANK='0')
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
    r = _run(['--gpus', '8', '--dry-launch'], RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    assert r.returncode != 0 and '--gpus 8' in r.stderr
    r = _run(['--gpus', '1', '--dry-launch'], RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')      # what the driver's launcher does at N = 1
    assert r.returncode == 0 and _json_lines(r.stdout)[0]['n_gpus'] == 1


def test_the_parent_module_imports_neither_torch_nor_the_hip_library():
    code = ('import sys; sys.argv = ["bench.py"]; import bench; '
            'assert "torch" not in sys.modules, "bench imports torch at module level"; '
            'assert "asr_amd._lib" not in sys.modules, "bench loads the HIP library at module level"; print("clean")')
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'clean' in r.stdout, r.stderr


def test_plan_calls_gives_every_handle_the_same_number_of_steps():
    sys.path.insert(0, ROOT)
    import bench
    for k in range(0, 41):
        for handles in (1, 2, 3, 4, 6):
            for per_call in (1, 2, 3, 4):
                calls = bench.plan_calls(k, handles, per_call)
                assert sum(calls) == k and all(1 <= c <= per_call + (1 if per_call >= 2 else 0) for c in calls)
                assert calls == sorted(calls, reverse=True)                   # the large calls first
                # every handle's share in calls of per_call steps; an odd last step joins the handle's last call
                share = [k // handles + (1 if i < k % handles else 0) for i in range(handles)]
                want = sum((s // per_call) + (1 if (s % per_call > 1 or (s % per_call == 1 and (s < per_call or per_call == 1))) else 0) for s in share)
                assert len(calls) == want, (k, handles, per_call, calls)
    assert bench.plan_calls(20, 4, 2) == [3] * 4 + [2] * 4
    assert bench.plan_calls(8, 4, 2) == [2] * 4
    assert bench.plan_calls(5, 4, 2) == [2, 1, 1, 1]
    assert bench.plan_calls(10, 4, 2) == [3, 3, 2, 2]
    assert bench.plan_calls(6, 4, 1) == [1] * 6
