"""One process per GPU, started by the benchmark itself when nobody else did.

``python bench.py --gpus N`` under torchrun finds RANK / WORLD_SIZE in the environment and runs as one rank.  Started plainly with
N > 1 it must not die on a missing launcher: ``spawn_ranks`` starts N CHILD processes of the same script (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), BEFORE the parent has touched the GPU -- a process that has initialised HIP
must never exec or fork into another GPU program on this pool -- relays rank 0's stdout (the one JSON line) and returns non-zero when
any rank failed."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
from typing import List, Sequence


def needs_spawn(gpus: int) -> bool:
    return gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), FMI_SELF_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes fails without it on this image
    env.setdefault("OMP_NUM_THREADS", "4")
    return env


def spawn_ranks(script: str, argv: Sequence[str], gpus: int, timeout_s: float = 3000.0) -> int:
    """returns the exit code for the parent: 0 only if every rank exited 0.  Rank 0's stdout is passed through line by line; the other
    ranks' stdout is dropped, every rank's stderr goes to the parent's stderr with a rank prefix."""
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(gpus):
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=rank_environment(r, gpus, port), stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=subprocess.PIPE, text=True))

    def pump(stream, sink, prefix):
        for line in stream:
            sink.write(prefix + line)
            sink.flush()

    threads = [threading.Thread(target=pump, args=(procs[0].stdout, sys.stdout, ""), daemon=True)]
    threads += [threading.Thread(target=pump, args=(p.stderr, sys.stderr, f"[rank {r}] "), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    codes = []
    try:
        for p in procs:
            codes.append(p.wait(timeout=timeout_s))
    except subprocess.TimeoutExpired:
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.kill()
        codes = [p.wait() for p in procs]
        sys.stderr.write(f"spawn_ranks: timeout after {timeout_s} s, ranks killed\n")
        return 124
    for t in threads:
        t.join(timeout=5)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("spawn_ranks: ranks failed (rank, exit code): %s\n" % bad)
        return 1
    return 0
