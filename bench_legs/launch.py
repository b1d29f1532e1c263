"""bench_legs.launch -- `python bench.py --gpus N` without torch.distributed.run: one rank per GPU, spawned here."""
import os
import sys
import time

import torch

from bench_legs.common import ROOT


def _flush_c_stdio():
    """RCCL printf()s a version banner when the first communicator is created; with stdout a pipe or a file it sits in C stdio's buffer until exit,
    i.e. it would land AFTER a line printed from Python"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                         # noqa: BLE001
        pass


def _free_port():
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    return port


def self_launch(n, argv, script=None):
    """`python bench.py --gpus N` without torch.distributed.run: start N ranks of this script (one per visible GPU, rendezvous on 127.0.0.1 at a free
    port), rank 0's stdout is this process's stdout (its JSON line stays the last thing written there), the other ranks' stdout goes to stderr.
    Returns the exit code: 0 only if every rank exited 0; a rank that dies takes the others down with it (exact PIDs, after a grace period) instead of
    leaving them in a collective forever."""
    backend = os.environ.get("AC_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev == 0:
        print("bench.py needs an MI355X (torch.cuda.is_available() is False); the hot path has no CPU fallback", file=sys.stderr)
        return 1
    if ndev < n and backend == "nccl":
        print(f"bench.py --gpus {n}: only {ndev} GPU(s) visible; RCCL needs one device per rank (AC_DIST_BACKEND=gloo runs the N > 1 code path with "
              f"ranks sharing a device -- a plumbing check, not a measurement)", file=sys.stderr)
        return 2
    # HSA_ENABLE_IPC_MODE_LEGACY: this pool's host driver supports dmabuf IPC only -- the image exports HSA_ENABLE_IPC_MODE_LEGACY=0 for that reason and
    # its documentation says RCCL / cross-process device memory fails with "hipIpcGetMemHandle: invalid argument" without it (the task environment's own
    # statement; no multi-GPU box was available to this build to observe either outcome).  So: an inherited value is passed through untouched; with none
    # inherited the ranks get 0, and if that job dies on an RCCL job (any rank non-zero) it is started ONCE more with the variable unset -- the line
    # records which attempt produced it (`hsa_ipc_mode_legacy`).  AC_BENCH_IPC_RETRY=0 switches the second attempt off.
    inherited = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    attempts = [(inherited, "inherited from the environment")] if inherited is not None else [("0", "launcher default (dmabuf IPC, as the image exports it)")]
    if inherited is None and backend == "nccl" and os.environ.get("AC_BENCH_IPC_RETRY", "1") != "0":
        attempts.append((None, "unset (second attempt: the first, with 0, failed)"))
    rc = 1
    for k, (ipc, why) in enumerate(attempts):
        rc = _launch_once(n, argv, script, ipc, f"attempt {k + 1}: {why}")
        if rc == 0:
            break
        if k + 1 < len(attempts):
            print(f"bench.py: the {n}-rank job failed (exit {rc}) with HSA_ENABLE_IPC_MODE_LEGACY={ipc}; one more attempt with it unset", file=sys.stderr)
    return rc


def _launch_once(n, argv, script, ipc, ipc_note):
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AC_BENCH_LAUNCHER="self", AC_BENCH_IPC_NOTE=ipc_note)
        if ipc is None:
            env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
        else:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = ipc
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, script or os.path.join(ROOT, "bench.py")] + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    rc, dead_since = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and dead_since is None:
            dead_since = time.time()
        if dead_since is not None and time.time() - dead_since > float(os.environ.get("AC_BENCH_GRACE_S", "20")):
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for r, p in enumerate(procs):
        if p.returncode != 0:
            print(f"bench.py: rank {r} exited with {p.returncode}", file=sys.stderr)
            rc = rc or (p.returncode if p.returncode and p.returncode > 0 else 1)
    return rc
