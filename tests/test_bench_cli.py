"""bench.py's launch path on the CPU: `--gpus N` without a launcher must start N ranks itself (round-1 defect:
the flag was parsed and ignored), and under a launcher (WORLD_SIZE set) every process is one rank."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    # stdout is the one JSON line and nothing else (gloo's "[Gloo] Rank 0 is connected ..." goes to stderr)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out
    return json.loads(lines[0])


def test_gpus_flag_spawns_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["dry_run"] is True


def test_under_a_launcher_each_process_is_one_rank():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    try:
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    assert all(p.returncode == 0 for p in procs), outs
    assert _json_line(outs[0][0])["n_gpus"] == 2
    assert not outs[1][0].strip()                                                # only rank 0 prints
