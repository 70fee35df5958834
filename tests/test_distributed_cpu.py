"""world_size-2 gloo tests (CPU) of the N>1 host logic: unit partitioning, the max-over-ranks timing contract of
bench.py, and the mpc-net style share exchange of an open."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    from czk_amd import parallel
    import orc
    from util import rand_fr_canonical
    r, w, _ = parallel.init("gloo")
    assert (r, w) == (rank, world)
    units = parallel.partition_units(5, world, rank)
    # each rank "takes" 0.1 s * (rank + 1): the job time is the slowest rank's
    rate, t = parallel.aggregate_throughput(float(len(units)), 0.1 * (rank + 1))
    # an SPDZ open: additive shares of a secret vector, one share per party (share/spdz.rs:150-185)
    n = 64
    secret = orc.fr_from_repr(rand_fr_canonical(7, n))
    share0 = orc.fr_from_repr(rand_fr_canonical(8, n))
    mine = share0 if rank == 0 else orc.fr_sub(secret, share0)
    gathered = parallel.all_gather_shares(torch.from_numpy(mine.view(np.int64)))
    total = gathered[0].numpy().view(np.uint64)
    for k in range(1, world):
        total = orc.fr_add(total, gathered[k].numpy().view(np.uint64))
    parallel.barrier()
    q.put((rank, units, rate, t, bool(np.array_equal(total, secret))))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_partition_timing_and_open():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]          # 5 units over 2 ranks
    for _, _, rate, t, ok in res:
        assert ok                                                   # opened value == secret on every party
        assert abs(t - 0.2) < 1e-9 and abs(rate - 5 / 0.2) < 1e-6   # all units / max-over-ranks time


def test_partition_units_covers_everything():
    sys.path.insert(0, ROOT)
    from czk_amd import parallel
    for n in (0, 1, 7, 8, 9):
        for w in (1, 2, 3, 8):
            parts = [parallel.partition_units(n, w, r) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


@pytest.mark.parametrize("gpus,layout", [(2, "replica"), (3, "party")])
def test_bench_launcher_starts_one_rank_per_gpu(gpus, layout):
    """`python bench.py --gpus N` (no torchrun environment) must start N ranks itself and report the world size the
    process group saw; --dry-run skips the GPU work so the launcher is covered on CPU (gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run", "--layout", layout, "--parties", str(gpus if layout == "party" else 2),
           "--no-multi-gpu-report"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] and d["n_gpus"] == gpus and d["ranks_seen_by_backend"] == gpus and "multi_gpu_report" not in d
    assert abs(d["max_over_ranks_s"] - 0.001 * gpus) < 1e-9      # the slowest rank's time


@pytest.mark.parametrize("gpus", [2, 3])
def test_multi_gpu_report_shape(gpus):
    """`bench.py --gpus N` on N > 1 GPUs (the driver's SCALE run) follows its replica line with the party / split layout children
    (bench.multi_gpu_report); --dry-run produces the report's shape on CPU: one launch per kind of child (one-GPU reference, party layout
    over N ranks, the Groth16 party layout under a real key with `proof_verifies`, split layout over N ranks), the rest as planned commands.  Every party child names the one-GPU run whose digest it must
    reproduce, covers both exchange patterns and both transports, and the config is the BASELINE one whose party count is N."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run"], capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                        # ONE JSON line, whatever the children print
    d = json.loads(lines[0])
    rep = d["multi_gpu_report"]
    assert d["n_gpus"] == gpus and d["ranks_seen_by_backend"] == gpus
    baseline_cfg = {2: "marlin_spdz2_2e20", 3: "plonk_gsz3_2e18"}[gpus]
    party = {k: v for k, v in rep.items() if "/party/" in k}
    assert {k.split("/", 2)[2] for k in party if k.startswith(baseline_cfg)} == {"torch/ring", "torch/p2p", "czk/ring", "czk/p2p", "czk-ipc/ring"}
    for k, v in party.items():
        assert v["reference"] == k.split("/")[0] + "/one_gpu" and v["reference"] in rep, k
        assert f"--gpus {gpus} --layout party" in v["command"] and f"--parties {gpus}" in v["command"]
    launched = {k: v for k, v in rep.items() if v.get("dry_run") is True}
    assert {k.split("/")[1] for k in launched} == {"one_gpu", "party", "split", "verify"}
    verify = {k: v for k, v in rep.items() if "/verify/" in k}
    assert verify and all(k.startswith("groth16") and "--real-key" in v["command"] and f"--gpus {gpus} --layout party" in v["command"] for k, v in verify.items())
    for k, v in launched.items():
        want = 1 if k.endswith("/one_gpu") else gpus
        assert v["n_gpus"] == want and v["ranks_seen_by_backend"] == want, (k, v)
    assert rep["groth16_spdz2_2e20/split"]["reference"] == "headline"
    assert not any("error" in v for v in rep.values()), rep


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_polyiop_party_digests_agree_between_layouts():
    """bench.py compares the party-per-rank layout of the Plonk / Marlin provers with the all-lanes-on-one-GPU layout party by
    party: the digest of party p's lane slice of the one-GPU outputs must equal the digest rank p computes over its own lanes;
    single-lane (public) entries belong to every party."""
    import importlib.util
    import os
    import numpy as np
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(5)
    parties, per = 2, 2
    lanes = parties * per
    full = {"w_cmt": (rng.integers(0, 1 << 63, size=(lanes, 12), dtype=np.uint64), np.zeros(lanes, dtype=np.uint8)),
            "t_cmt": (rng.integers(0, 1 << 63, size=(1, 12), dtype=np.uint64), np.zeros(1, dtype=np.uint8)),       # a public polynomial
            "evals_beta": [rng.integers(0, 1 << 63, size=(lanes, 4), dtype=np.uint64), rng.integers(0, 1 << 63, size=(1, 4), dtype=np.uint64)],
            "open_beta": {"value": rng.integers(0, 1 << 63, size=(lanes, 4), dtype=np.uint64), "point": 12345, "fold": 7,
                          "proof": (rng.integers(0, 1 << 63, size=(lanes, 12), dtype=np.uint64), np.zeros(lanes, dtype=np.uint8))}}

    def mine(x, p):
        if isinstance(x, dict):
            return {k: mine(v, p) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(mine(v, p) for v in x)
        if isinstance(x, np.ndarray) and x.shape[0] == lanes:
            return x[p * per:(p + 1) * per].copy()
        return x
    replica = bench._polyiop_party_digests(full, lanes, parties)
    ranks = [bench._polyiop_party_digests(mine(full, p), per, 1, only=0)[0] for p in range(parties)]
    assert replica == ranks and replica[0] != replica[1]
