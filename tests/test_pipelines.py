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


@pytest.mark.parametrize("n_gates,parties", [(8, 3), (64, 3)])
def test_plonk_gsz_pipeline_matches_checker(orc, n_gates, parties):
    """mpc-plonk's Prover::prove, GSZ shares of 3 parties as 3 lanes on one GPU (the 1-GPU form of configs[2])."""
    import czk_amd
    from czk_amd import polyvm
    from oracle_backend import make_backend
    ctx = polyvm.shared_stream_context(czk_amd)
    md = polyvm.plonk_max_degree(n_gates)
    gpu = polyvm.GpuBackend(czk_amd, ctx, parties, md)
    cpu = make_backend(orc, polyvm, parties, md, bases=gpu.bases_host())
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
    cpu = make_backend(orc, polyvm, 4, md, lift=lift, bases=gpu.bases_host())
    from oracle_backend import make_lockstep
    ls = make_lockstep(polyvm, gpu, cpu)
    polyvm.marlin_prove(ls, polyvm.marlin_inputs(ls, n_constraints))
    got = polyvm.marlin_prove(gpu, polyvm.marlin_inputs(gpu, n_constraints))
    want = polyvm.marlin_prove(cpu, polyvm.marlin_inputs(cpu, n_constraints))
    _compare(got, want)
    ctx.close()
