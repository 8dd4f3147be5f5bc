"""PWD Jensen-Shannon metric.

CPU part: the numpy oracle against the golden vectors recorded from the reference's own
evaluate/evaluators.py (tests/golden/make_golden_pwd.py), incl. the reference's saved chignolin
histograms.  GPU part (-m gpu): the HIP kernels through the C ABI, bit-exact against the oracle and
the golden vectors (counts are integers), the PwdEvaluator mirror's JS equal to the reference's, and
size-independent properties at BASELINE config sizes.
"""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import pwd_metric as om
from oracle import synth

CASES = ["n10_off3", "n5_off1", "n6_off0", "n35_off3"]


def unflat(lens, flat):
    out, o = [], 0
    for n in lens:
        out.append(np.asarray(flat[o:o + n]))
        o += n
    return out


# ------------------------------------------------------------------ CPU: oracle vs reference vectors
@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_vectors(golden, tag):
    g = golden(f"pwd_synth_{tag}.npz")
    off = int(g["offset"])
    assert np.array_equal(om.pair_max(g["x_val"], off), g["gt_max"])
    nb = om.nbins_for(g["gt_max"])
    assert list(nb) == list(g["gt_hist_len"])
    for a, b in zip(om.histograms(g["x_val"], off, nb), unflat(g["gt_hist_len"], g["gt_hist"])):
        assert np.array_equal(a, b.astype(np.int64))
    js, per = om.js_divergence_pwd(unflat(g["gt_hist_len"], g["gt_hist"]), g["x_samp"], g["gt_max"], off)
    for a, b in zip(per, unflat(g["samp_hist_len"], g["samp_hist"])):
        assert np.array_equal(a, b)
    assert js == float(g["js"])          # same numpy reductions as the reference: bit-equal


def test_oracle_on_reference_saved_histograms(golden):
    g = golden("pwd_chignolin_ref.npz")
    js, _ = om.js_divergence_pwd(unflat(g["gt_hist_len"], g["gt_hist"]), g["x_samp"], g["gt_max"], 3)
    assert js == float(g["js"])
    assert len(g["gt_max"]) == 28 and abs(g["gt_hist"].sum() / 28 - 106948) < 1    # 28 pairs x 106948 frames


def test_histc_edge_semantics():
    """values on bin edges, on max, above max, and the float32 quirks torch.histc has."""
    nb = 7
    v = np.array([0.0, 0.1, 0.0999999, 0.7, 0.70000005, 0.69999999, 0.35, -0.0, 0.8], np.float32)
    ref = torch.histc(torch.from_numpy(v), bins=nb, min=0, max=0.1 * nb).numpy().astype(np.int64)
    assert np.array_equal(om.histc(v, nb, 0.1 * nb), ref)
    rng = np.random.default_rng(3)
    for nb in (1, 2, 111, 296, 1000):
        v = (rng.random(20000) * 0.1 * nb * 1.01).astype(np.float32)
        ref = torch.histc(torch.from_numpy(v), bins=nb, min=0, max=0.1 * nb).numpy().astype(np.int64)
        assert np.array_equal(om.histc(v, nb, 0.1 * nb), ref)


def test_pair_order_and_counts():
    import dff_amd
    for N, off in [(10, 3), (5, 1), (6, 0), (56, 3), (35, 3), (4, 3), (3, 5)]:
        i, j = om.triu_pairs(N, off)
        ti = torch.triu_indices(N, N, offset=off)
        assert np.array_equal(i, ti[0].numpy()) and np.array_equal(j, ti[1].numpy())
        assert dff_amd.binding.pwd_num_pairs(N, off) == len(i)


# ------------------------------------------------------------------ GPU: HIP kernels vs oracle / vectors
gpu = pytest.mark.gpu


def _hip_hists(x, off, nb):
    from dff_amd import binding
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    hmax = torch.tensor([0.1 * int(b) for b in nb], dtype=torch.float64).float()
    c = binding.pwd_hist(xd, off, torch.as_tensor(np.asarray(nb), dtype=torch.int32), hmax).cpu().numpy()
    return [c[p, :int(b)].astype(np.int64) for p, b in enumerate(nb)], c


@gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_max_and_hist_bit_exact(golden, tag):
    from dff_amd import binding
    g = golden(f"pwd_synth_{tag}.npz")
    off = int(g["offset"])
    xd = torch.from_numpy(g["x_val"]).cuda()
    mx = binding.pwd_max(xd, off).cpu().numpy()
    assert np.array_equal(mx, g["gt_max"])
    nb = om.nbins_for(mx)
    hs, raw = _hip_hists(g["x_val"], off, nb)
    for p, (a, b) in enumerate(zip(hs, unflat(g["gt_hist_len"], g["gt_hist"]))):
        assert np.array_equal(a, b.astype(np.int64)), p
        assert raw[p, len(a):].sum() == 0            # nothing beyond the pair's own bins


@gpu
@pytest.mark.parametrize("tag", CASES)
def test_evaluator_js_equals_reference(golden, tag, tmp_path):
    from dff_amd.evaluate import PwdEvaluator
    g = golden(f"pwd_synth_{tag}.npz")
    off = int(g["offset"])
    ref_file = str(tmp_path / "ref.pickle")
    e = PwdEvaluator(torch.from_numpy(g["x_val"]), mol_name="synth", offset=off, saved_ref=ref_file)
    assert np.array_equal(e.gt_max.numpy(), g["gt_max"])
    assert float(e.eval(torch.from_numpy(g["x_samp"]))) == float(g["js"])
    # the pickle it writes has the reference's layout and is read back like the reference reads it
    d = pickle.load(open(ref_file, "rb"))
    assert set(d) == {"gt_max", "gt_hist"} and d["gt_hist"][0].dtype == torch.float32
    e2 = PwdEvaluator(None, mol_name="synth", offset=off, saved_ref=ref_file)
    assert float(e2.eval(torch.from_numpy(g["x_samp"]).cuda())) == float(g["js"])


@gpu
def test_evaluator_on_reference_saved_histograms(golden, tmp_path):
    """gt from the reference's saved_pwd_CHIGNOLIN_testset_offset_3 data, samples on the GPU."""
    from dff_amd.evaluate import PwdEvaluator
    g = golden("pwd_chignolin_ref.npz")
    ref_file = str(tmp_path / "saved_pwd_CHIGNOLIN_testset_offset_3.pickle")
    hists = [torch.from_numpy(h.copy()) for h in unflat(g["gt_hist_len"], g["gt_hist"])]
    pickle.dump({"gt_max": torch.from_numpy(g["gt_max"]), "gt_hist": hists}, open(ref_file, "wb"))
    e = PwdEvaluator(None, mol_name="chignolin", offset=3, saved_ref=ref_file)
    assert float(e.eval(torch.from_numpy(g["x_samp"]))) == float(g["js"])


@gpu
@pytest.mark.parametrize("N,n,off", [(10, 1, 3), (10, 63, 3), (10, 65, 3), (10, 4097, 3), (56, 333, 3), (35, 1000, 1),
                                     (64, 130, 0), (2, 500, 1)])
def test_hip_vs_oracle_ragged_sizes(N, n, off):
    """tile tails, one structure, maximum bead count, misaligned sizes: bit-exact against the oracle."""
    from dff_amd import binding
    x = (synth.normal((n, N, 3), 99, N * 1000 + n) * 3.0).astype(np.float32)
    mx = binding.pwd_max(torch.from_numpy(x).cuda(), off).cpu().numpy()
    assert np.array_equal(mx, om.pair_max(x, off))
    nb = om.nbins_for(mx)
    hs, _ = _hip_hists(x, off, nb)
    for a, b in zip(hs, om.histograms(x, off, nb)):
        assert np.array_equal(a, b)


@gpu
def test_hip_edge_cases():
    from dff_amd import binding
    # empty input: zeros, no launch
    x0 = torch.zeros((0, 10, 3), device="cuda")
    assert binding.pwd_max(x0, 3).abs().sum().item() == 0
    h = binding.pwd_hist(x0, 3, torch.full((28,), 5, dtype=torch.int32), torch.full((28,), 0.5))
    assert h.shape == (28, 5) and h.sum().item() == 0
    # identical structures: every count in one bin; distances beyond hmax are dropped like histc does
    x = torch.from_numpy((synth.normal((1, 10, 3), 5, 5) * 2).astype(np.float32)).repeat(777, 1, 1).cuda()
    nb = torch.full((28,), 10, dtype=torch.int32)
    h = binding.pwd_hist(x, 3, nb, torch.full((28,), 1.0)).cpu().numpy()
    d = om.pwd_triu(x[:1].cpu().numpy(), 3)[0]
    for p in range(28):
        assert h[p].sum() == (777 if d[p] <= 1.0 else 0)
    # a view that is not 16-byte aligned takes the scalar load path
    big = torch.from_numpy((synth.normal((501, 10, 3), 6, 6) * 3).astype(np.float32)).cuda()
    flat = torch.empty(501 * 30 + 1, device="cuda")
    flat[1:] = big.reshape(-1)
    xv = flat[1:].view(501, 10, 3)
    assert xv.data_ptr() % 16 != 0
    assert torch.equal(binding.pwd_max(xv, 3), binding.pwd_max(big, 3))
    # bad arguments are errors, not crashes
    with pytest.raises(ValueError):
        binding.pwd_max(big.cpu(), 3)
    with pytest.raises(ValueError):
        binding.pwd_hist(big, 3, torch.ones(5, dtype=torch.int32), torch.ones(5))
    with pytest.raises(ValueError):
        binding.pwd_max(big, 10)        # no pairs at that offset


@gpu
@pytest.mark.parametrize("cfg,N,n", [("chignolin config 2", 10, 10240), ("villin config 4 / 8 GPUs", 35, 102400),
                                     ("protein G config 5 / 8 GPUs", 56, 51200)])
def test_full_size_properties(cfg, N, n):
    """At BASELINE.json's output sizes: every structure lands in exactly one bin of every pair, the
    histogram is additive over a split of the structures and invariant under their permutation, and a
    strided sub-sample agrees with the oracle bit for bit."""
    from dff_amd import binding
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((n, N, 3), device="cuda", generator=g) * 5.0
    mx = binding.pwd_max(x, 3)
    nb = (torch.div(mx.cpu(), 0.1, rounding_mode="floor") + 1).to(torch.int32)
    hmax = torch.tensor([0.1 * int(b) for b in nb], dtype=torch.float64).float()
    h = binding.pwd_hist(x, 3, nb, hmax)
    assert torch.all(h.sum(1) == n)
    cut = n // 3 + 5
    h1 = binding.pwd_hist(x[:cut].contiguous(), 3, nb, hmax)
    h2 = binding.pwd_hist(x[cut:].contiguous(), 3, nb, hmax)
    assert torch.equal(h1 + h2, h)
    perm = torch.randperm(n, device="cuda", generator=g)
    assert torch.equal(binding.pwd_hist(x[perm].contiguous(), 3, nb, hmax), h)
    assert torch.equal(torch.maximum(binding.pwd_max(x[:cut].contiguous(), 3), binding.pwd_max(x[cut:].contiguous(), 3)), mx)
    sub = x[::101].contiguous()
    hs = binding.pwd_hist(sub, 3, nb, hmax).cpu().numpy()
    ref = om.histograms(sub.cpu().numpy(), 3, nb.numpy())
    for p, r in enumerate(ref):
        assert np.array_equal(hs[p, :len(r)], r)
