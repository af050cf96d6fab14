import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = {k: v for k, v in np.load(os.path.join(GOLDEN, name)).items()}
        return cache[name]
    return load


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


# ---- measured parity errors: tests call record_parity(group, name, value); the table is written at session end to
# gpurun_out/parity_errors.json (copied into profiles/ per round): tolerances in the tests are set against these numbers
_PARITY = {}


def record_parity(group, name, value):
    _PARITY.setdefault(group, {})[name] = float(value)


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_errors.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(_PARITY)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)


def spawn_ranks(worker, world, extra_args=(), deadline=600):
    """start `world` processes `worker(rank, world, port, queue, *extra_args)` on a free rendezvous port and collect one queue item per rank.  Up to three attempts:
    the port is picked, released and bound again by rank 0 a moment later -- on a busy machine another process can take it in between (EADDRINUSE).  A worker that
    dies BEFORE reporting is retried on a fresh port; one that reports is not (its checks are the test's business)."""
    import multiprocessing as mp
    import queue as _queue
    import socket
    import time
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra_args)) for r in range(world)]
        for p in procs:
            p.start()
        res, t0, died = [], time.time(), False
        while len(res) < world:
            try:
                res.append(q.get(timeout=5))
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    died = True
                    break
                assert time.time() - t0 < deadline, "workers neither reported nor exited in %d s" % deadline
        if died:
            for p in procs:
                p.join(timeout=30)
                if p.is_alive():
                    p.kill()
            assert attempt < 2, "a worker died before reporting, three times in a row"
            continue
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        return res
