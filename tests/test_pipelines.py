"""Plonk and Marlin local-compute pipelines (BASELINE configs[2] and configs[3] at small sizes): the GPU path
(collaborative-zksnark_amd/polyvm.py GpuBackend, through the C ABI) against the same prover sequence executed on the CPU checker
(tests/oracle_backend.py) -- every commitment and every opening (value and proof) must agree."""
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
    assert one.returncode == 0, one.stderr[-2000:]
    many = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(parties), "--layout", "party", "--backend", "gloo", "--device", "0"]
                          + common, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert many.returncode == 0, many.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    d2 = json.loads(many.stdout.strip().splitlines()[-1])
    assert d2["n_gpus"] == parties and d2["ranks_seen_by_backend"] == parties and d2["config"]["layout"] == "party"
    assert d1["results_checked"] and d2["results_checked"]
    assert "batches per proof" in d2["config"]["workload"] and ", 0 batches" not in d2["config"]["workload"]
    assert d1["config"]["results_sha256"] == d2["config"]["results_sha256"]
