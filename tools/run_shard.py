"""Start N ranks of tools/shard_rank.py (fresh processes, gloo on 127.0.0.1) and relay rank 0's JSON line.
This parent never touches the GPU.   python tools/run_shard.py N [--captures ... --seconds ... as tools/shard_rank.py]"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, extra, timeout=1100):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "shard_rank.py")] + list(extra), env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        out = procs[0].communicate(timeout=timeout)[0]
        rcs = [p.wait(timeout=timeout) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return rcs, out


if __name__ == "__main__":
    n = int(sys.argv[1])
    rcs, out = run(n, sys.argv[2:])
    sys.stdout.write(out)
    sys.exit(max(abs(rc) for rc in rcs))
