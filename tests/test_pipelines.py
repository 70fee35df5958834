"""Plonk and Marlin local-compute pipelines (BASELINE configs[2] and configs[3] at small sizes): the GPU path
(collaborative-zksnark_amd/polyvm.py GpuBackend, through the C ABI) against the same prover sequence executed on the CPU checker
(tests/oracle_backend.py) -- every commitment and every opening (value and proof) must agree."""
import os

import numpy as np
import pytest

from util import rand_fr_canonical

pytestmark = pytest.mark.gpu


def _compare(got, want):
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if k.endswith("_cmt"):
            assert np.array_equal(got[k][1], want[k][1]), k
            assert np.array_equal(got[k][0][got[k][1] == 0], want[k][0][want[k][1] == 0]), k
        elif k.startswith("evals_"):
            assert len(got[k]) == len(want[k]) and all(np.array_equal(a, b) for a, b in zip(got[k], want[k])), k
        else:
            assert np.array_equal(got[k]["value"], want[k]["value"]), k
            assert np.array_equal(got[k]["proof"][1], want[k]["proof"][1]), k
            assert np.array_equal(got[k]["proof"][0], want[k]["proof"][0]), k
            assert ("random_v" in got[k]) == ("random_v" in want[k]) and np.array_equal(got[k].get("random_v"), want[k].get("random_v")), k


@pytest.mark.parametrize("n_gates,parties", [(8, 3), (64, 3)])
def test_plonk_gsz_pipeline_matches_checker(orc, n_gates, parties):
    """mpc-plonk's Prover::prove, GSZ shares of 3 parties as 3 lanes on one GPU (the 1-GPU form of configs[2])."""
    import czk_amd
    from czk_amd import polyvm
    from oracle_backend import make_backend
    ctx = polyvm.shared_stream_context(czk_amd)
    md = polyvm.plonk_max_degree(n_gates)
    gpu = polyvm.GpuBackend(czk_amd, ctx, parties, md)
    cpu = make_backend(orc, polyvm, parties, md, bases=gpu.bases_host(), bases_gamma=gpu.bases_gamma_host())
    from oracle_backend import make_lockstep
    ls = make_lockstep(polyvm, gpu, cpu)
    polyvm.plonk_prove(ls, polyvm.plonk_inputs(ls, n_gates))   # operation by operation: a divergence names the primitive
    gpu.msm_count = 0
    got = polyvm.plonk_prove(gpu, polyvm.plonk_inputs(gpu, n_gates))
    want = polyvm.plonk_prove(cpu, polyvm.plonk_inputs(cpu, n_gates))
    _compare(got, want)
    assert gpu.msm_count == parties * (7 + 16 - 2) + 2          # 7 commitments + 16 openings; the two openings of public polynomials are single-lane
    ctx.close()


@pytest.mark.parametrize("n_constraints", [16, 100])
def test_marlin_spdz_pipeline_matches_checker(orc, n_constraints):
    """Marlin's AHP prover rounds + commitments + openings, SPDZ share lanes of 2 parties (4 lanes; public data added on the
    king's lanes only) -- the 1-GPU form of configs[3]."""
    import czk_amd
    from czk_amd import polyvm
    from oracle_backend import make_backend
    ctx = polyvm.shared_stream_context(czk_amd)
    md = polyvm.marlin_max_degree(n_constraints)
    lift = (1, 1, 0, 0)
    gpu = polyvm.GpuBackend(czk_amd, ctx, 4, md, lift=lift)
    cpu = make_backend(orc, polyvm, 4, md, lift=lift, bases=gpu.bases_host(), bases_gamma=gpu.bases_gamma_host())
    from oracle_backend import make_lockstep
    ls = make_lockstep(polyvm, gpu, cpu)
    polyvm.marlin_prove(ls, polyvm.marlin_inputs(ls, n_constraints))
    got = polyvm.marlin_prove(gpu, polyvm.marlin_inputs(gpu, n_constraints))
    want = polyvm.marlin_prove(cpu, polyvm.marlin_inputs(cpu, n_constraints))
    _compare(got, want)
    ctx.close()


@pytest.mark.parametrize("workload,parties,constraints", [("plonk", 3, 64), ("marlin", 2, 100)])
def test_party_per_rank_layout_of_the_polynomial_provers_matches_the_one_gpu_layout(workload, parties, constraints):
    """bench.py --workload plonk|marlin --layout party: one party per rank (its 1 GSZ lane / 2 SPDZ lanes), every batch of
    evaluations between two challenges opened over torch.distributed (GSZ batch_open / SPDZ two-round batch_open with the MAC
    check).  Party by party, commitments, evaluations and opening proofs must equal the layout with all parties' lanes on one
    GPU; the ranks share this box's GPU, so the exchange runs over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--workload", workload, "--parties", str(parties), "--constraints", str(constraints), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert one.returncode == 0, __import__('util').child_errors(one.stderr)
    many = __import__('util').run_ranks([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(parties), "--layout", "party", "--backend", "gloo", "--device", "0"]
                          + common, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert many.returncode == 0, __import__('util').child_errors(many.stderr)
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    d2 = json.loads(many.stdout.strip().splitlines()[-1])
    assert d2["n_gpus"] == parties and d2["ranks_seen_by_backend"] == parties and d2["config"]["layout"] == "party"
    assert d1["results_checked"] and d2["results_checked"]
    assert "batches per proof" in d2["config"]["workload"] and ", 0 batches" not in d2["config"]["workload"]
    assert d1["config"]["results_sha256"] == d2["config"]["results_sha256"]
    # the same layout with the opens through the library's own communicator (czk_net, shared-memory transport: parallel.use_net)
    czkn = __import__('util').run_ranks([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(parties), "--layout", "party", "--backend", "gloo", "--device", "0", "--net", "czk"]
                          + common, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert czkn.returncode == 0, __import__('util').child_errors(czkn.stderr)
    d3 = json.loads(czkn.stdout.strip().splitlines()[-1])
    assert d3["net"] == "czk_net shm" and d3["results_checked"] and d3["config"]["results_sha256"] == d1["config"]["results_sha256"]
    assert "batches per proof" in d3["config"]["workload"] and ", 0 batches" not in d3["config"]["workload"]


def _load_cpp_dump(path):
    """tools/polyvm_host.hpp dump_output -> the structure polyvm.*_prove returns (commitments as (aff (L, 12), inf (L,)), values as (L, 4))"""
    import json
    raw = json.load(open(path))

    def cmt(v):
        b = [bytes.fromhex(s) for s in v]
        return np.stack([np.frombuffer(x[:96], dtype=np.uint64) for x in b]), np.array([x[96] for x in b], dtype=np.uint8)

    def val(v):
        return np.stack([np.frombuffer(bytes.fromhex(s), dtype=np.uint64) for s in v])
    out = {}
    for k, v in raw.items():
        if k.endswith("_cmt"):
            out[k] = cmt(v)
        elif k.startswith("evals_"):
            out[k] = [val(x) for x in v]
        else:
            o = {"value": val(v["value"]), "point": v["point"], "proof": cmt(v["proof"]), "of": v["of"]}
            if "random_v" in v:
                o["random_v"] = val(v["random_v"])
            if "terms" in v:
                o["terms"] = v["terms"]
            out[k] = o
    return out


@pytest.mark.parametrize("workload,size,parties", [("plonk", 64, 3), ("marlin", 100, 2), ("plonk", 1 << 18, 3), ("marlin", 1 << 20, 2)])
def test_cpp_polynomial_machine_host_matches_checker(orc, tmp_path, workload, size, parties):
    """tools/host_demo.cpp plonk | marlin (tools/polyvm_host.hpp: the C++ twin of polyvm.py over include/czk.h -- no torch, no Python, every
    array in one device arena, two proofs in flight on two contexts): every commitment, evaluation and opening (value, proof, blinding
    evaluation, point, combination terms) of its last proof against the SAME prover sequence executed on the CPU checker
    (tests/oracle_backend.py) -- the comparison the Python host passes in the tests above.  At the BASELINE sizes (configs[2]: 2^18 gates,
    configs[3]: 2^20 constraints) the checker-backed machine is out of reach; there the two product hosts are compared with each other."""
    import subprocess
    import czk_amd
    from czk_amd import polyvm
    from oracle_backend import make_backend
    from test_abi import _build_host_demo
    dump = str(tmp_path / "pvm.json")
    r = subprocess.run([_build_host_demo(), workload, "--constraints", str(size), "--parties", str(parties), "--steps", "3", "--warmup", "1", "--inflight", "2",
                        "--dump", dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '"in_flight_provers_equal": true' in r.stdout, r.stdout + r.stderr
    got = _load_cpp_dump(dump)
    ctx = polyvm.shared_stream_context(czk_amd)
    if workload == "plonk":
        lanes, lift, md = parties, None, polyvm.plonk_max_degree(size)
    else:
        lanes, lift, md = 2 * parties, tuple([1, 1] + [0] * (2 * parties - 2)), polyvm.marlin_max_degree(size)
    gpu = polyvm.GpuBackend(czk_amd, ctx, lanes, md, lift=lift)           # small sizes: only for the SRS the checker-backed machine commits under
    cpu = make_backend(orc, polyvm, lanes, md, lift=lift, bases=gpu.bases_host(), bases_gamma=gpu.bases_gamma_host()) if size <= 4096 else gpu
    want = polyvm.plonk_prove(cpu, polyvm.plonk_inputs(cpu, size)) if workload == "plonk" else polyvm.marlin_prove(cpu, polyvm.marlin_inputs(cpu, size))
    assert set(got) == set(want)
    for k in want:
        if k.endswith("_cmt"):
            assert np.array_equal(got[k][1], want[k][1]), k
            assert np.array_equal(got[k][0][got[k][1] == 0], want[k][0][want[k][1] == 0]), k
        elif k.startswith("evals_"):
            assert len(got[k]) == len(want[k]) and all(np.array_equal(a, b) for a, b in zip(got[k], want[k])), k
        else:
            assert np.array_equal(got[k]["value"], want[k]["value"]), k
            assert np.array_equal(got[k]["proof"][1], want[k]["proof"][1]) and np.array_equal(got[k]["proof"][0], want[k]["proof"][0]), k
            assert ("random_v" in got[k]) == ("random_v" in want[k]) and np.array_equal(got[k].get("random_v"), want[k].get("random_v")), k
            assert got[k]["point"] == polyvm.mont(want[k]["point"]).tobytes().hex(), k
            assert got[k]["of"] == want[k].get("of"), k
            if "terms" in want[k]:
                assert [(polyvm.mont(c).tobytes().hex(), n) for c, n in want[k]["terms"]] == [tuple(t) for t in got[k]["terms"]], k
    ctx.close()


@pytest.mark.parametrize("workload,size,world", [("plonk", 64, 3), ("marlin", 100, 2)])
def test_cpp_polynomial_machine_party_layout_matches_one_process(tmp_path, workload, size, world):
    """configs[2] / configs[3] in the reference's own layout from a compiled host: one process per party (tools/host_demo.cpp plonk | marlin
    --world N; the N processes share this box's GPU through the shared-memory transport of czk_net), each holding its party's lanes, every batch of
    evaluations opened through czk::Net (GszFieldShare / SpdzFieldShare::batch_open).  Party p's commitments, evaluations and opening proofs must be
    lanes [p * k, (p + 1) * k) of the one-process run; every party must see the same opened values, and they must be what the shares open to."""
    import subprocess
    from czk_amd import polyvm
    from test_abi import _build_host_demo
    exe = _build_host_demo()
    one = str(tmp_path / "one.json")
    r = subprocess.run([exe, workload, "--constraints", str(size), "--parties", str(world), "--steps", "2", "--warmup", "1", "--inflight", "1", "--dump", one],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    want = _load_cpp_dump(one)
    base = str(tmp_path / "party.json")
    r = __import__("util").run_ranks([exe, workload, "--constraints", str(size), "--world", str(world), "--steps", "2", "--warmup", "1", "--dump", base],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and '"layout": "party' in r.stdout, r.stdout + r.stderr
    import json
    k = 1 if workload == "plonk" else 2
    raws = [json.load(open(f"{base}.rank{p}")) for p in range(world)]
    opened = [raw.pop("opened") for raw in raws]
    assert all(o == opened[0] for o in opened) and opened[0]                  # every party saw the same opened vectors
    share_values = []                                                           # rank 0's view of every evaluation of a share polynomial, per party
    for p in range(world):
        tmp = str(tmp_path / f"r{p}.json")
        json.dump(raws[p], open(tmp, "w"))
        got = _load_cpp_dump(tmp)
        assert set(got) == set(want)
        vals = []
        for key in want:
            def lanes_of(a):
                return a[p * k:(p + 1) * k] if a.shape[0] == world * k else a
            if key.endswith("_cmt"):
                assert np.array_equal(got[key][1], lanes_of(want[key][1])) and np.array_equal(got[key][0], lanes_of(want[key][0])), (p, key)
            elif key.startswith("evals_"):
                assert all(np.array_equal(a, lanes_of(b)) for a, b in zip(got[key], want[key])), (p, key)
                vals += [a for a in got[key] if a.shape[0] == k and (k == 2)]
            else:
                assert np.array_equal(got[key]["value"], lanes_of(want[key]["value"])) and got[key]["point"] == want[key]["point"], (p, key)
                assert np.array_equal(got[key]["proof"][0], lanes_of(want[key]["proof"][0])) and np.array_equal(got[key]["proof"][1], lanes_of(want[key]["proof"][1])), (p, key)
                share_poly = got[key]["of"] is not None if workload == "plonk" else got[key]["value"].shape[0] == k
                if share_poly:
                    vals.append(got[key]["value"])
                if "random_v" in want[key]:
                    assert np.array_equal(got[key]["random_v"], lanes_of(want[key]["random_v"])), (p, key)
                    vals.append(got[key]["random_v"])
        share_values.append(vals)
    # what the shares open to: GSZ stand-in sharing -- every party holds the value itself, p(0) of the constant polynomial; SPDZ -- the sum of the sh lanes
    if workload == "plonk":
        expect = sorted(v[0].tobytes().hex() for v in share_values[0])
    else:
        expect = sorted(polyvm.mont(sum(polyvm.unmont(share_values[p][i][0]) for p in range(world))).tobytes().hex() for i in range(len(share_values[0])))
    assert sorted(h for batch in opened[0] for h in batch) == expect
