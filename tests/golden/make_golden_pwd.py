#!/usr/bin/env python3
"""Generate the golden vectors of the PWD Jensen-Shannon metric by running the REFERENCE.

Run once in the build container (the only place /root/reference exists):

    cd /tmp && python /root/repo/tests/golden/make_golden_pwd.py

Imports the reference's evaluate/evaluators.py (mdtraj, deeptime and its dataset module stubbed:
nothing on this path uses them) and records, as DATA only:
  pwd_synth_<tag>.npz   seeded structures -> PwdEvaluator(val).gt_max / gt_hist, the per-pair
                        torch.histc of a second seeded set, and PwdEvaluator.eval() of that set
  pwd_chignolin_ref.npz the reference's own saved histograms
                        evaluate/saved_references/saved_pwd_CHIGNOLIN_testset_offset_3.pickle
                        (a data file of the reference) + the reference's JS of a seeded set against it
It also checks oracle/pwd_metric.py against the reference on every vector.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {})


for name in ["mdtraj", "deeptime", "deeptime.decomposition", "datasets", "datasets.dataset_utils_empty"]:
    sys.modules[name] = _Stub(name)
import matplotlib  # noqa: E402

matplotlib.use("Agg")
sys.path.insert(0, REF)
sys.path.insert(1, REPO)
import evaluate.evaluators as ev  # noqa: E402  (reference)

from oracle import pwd_metric as om  # noqa: E402
from oracle import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def flat(hists):
    lens = np.array([len(h) for h in hists], np.int64)
    return lens, np.concatenate([np.asarray(h, np.float32) for h in hists])


def check_oracle(tag, x_val, x_samp, offset, gt_max, gt_hist, samp_hist, js):
    assert np.array_equal(om.pair_max(x_val, offset), gt_max), tag
    nb = om.nbins_for(gt_max)
    assert [len(h) for h in gt_hist] == list(nb), tag
    for a, b in zip(om.histograms(x_val, offset, nb), gt_hist):
        assert np.array_equal(a, np.asarray(b).astype(np.int64)), tag
    js_o, per = om.js_divergence_pwd([np.asarray(h, np.float32) for h in gt_hist], x_samp, gt_max, offset)
    for a, b in zip(per, samp_hist):
        assert np.array_equal(a, np.asarray(b, np.float32)), tag
    assert js_o == js, (tag, js_o, js)
    print(f"  oracle == reference on {tag}: JS = {js:.12f}")


def synth_case(tag, N, n_val, n_samp, offset, scale, stream):
    x_val = (synth.normal((n_val, N, 3), 4242, stream) * scale).astype(np.float32)
    x_samp = (synth.normal((n_samp, N, 3), 4242, stream + 1) * (scale * 1.07)).astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        e = ev.PwdEvaluator(torch.from_numpy(x_val), mol_name="synth", offset=offset,
                            saved_ref=os.path.join(td, "ref.pickle"))
        js = float(e.eval(torch.from_numpy(x_samp)))
        gt_max = e.gt_max.numpy()
        gt_hist = [h.numpy() for h in e.gt_hist]
        # the per-pair histograms js_divergence_pwd builds internally (evaluators.py:258-263)
        pwd = ev.get_pwd_triu_batch(torch.from_numpy(x_samp), offset)
        samp_hist = []
        for p, (gtm, col) in enumerate(zip(e.gt_max, pwd.t())):
            maxval = max(gtm, col.max())
            nb = int(torch.div(maxval, e.resolution, rounding_mode="floor") + 1)
            samp_hist.append(torch.histc(col, bins=nb, min=0, max=e.resolution * nb).numpy())
    check_oracle(tag, x_val, x_samp, offset, gt_max, gt_hist, samp_hist, js)
    gl, gf = flat(gt_hist)
    sl, sf = flat(samp_hist)
    np.savez_compressed(os.path.join(OUT, f"pwd_synth_{tag}.npz"), x_val=x_val, x_samp=x_samp, offset=offset,
                        gt_max=gt_max, gt_hist_len=gl, gt_hist=gf, samp_hist_len=sl, samp_hist=sf, js=js)


def chignolin_ref():
    src = os.path.join(REF, "evaluate", "saved_references", "saved_pwd_CHIGNOLIN_testset_offset_3.pickle")
    with open(src, "rb") as f:
        data = pickle.load(f)
    x_samp = (synth.normal((1000, 10, 3), 4242, 77) * 4.0).astype(np.float32)
    e = ev.PwdEvaluator(None, mol_name="chignolin", offset=3, saved_ref=src)
    js = float(e.eval(torch.from_numpy(x_samp)))
    gt_hist = [h.numpy() for h in data["gt_hist"]]
    js_o, _ = om.js_divergence_pwd(gt_hist, x_samp, data["gt_max"].numpy(), 3)
    assert js_o == js, (js_o, js)
    print(f"  oracle == reference on chignolin saved reference: JS = {js:.12f}")
    gl, gf = flat(gt_hist)
    np.savez_compressed(os.path.join(OUT, "pwd_chignolin_ref.npz"), x_samp=x_samp, offset=3,
                        gt_max=data["gt_max"].numpy(), gt_hist_len=gl, gt_hist=gf, js=js)


if __name__ == "__main__":
    synth_case("n10_off3", 10, 800, 600, 3, 4.0, 10)
    synth_case("n5_off1", 5, 500, 333, 1, 1.5, 20)
    synth_case("n6_off0", 6, 300, 200, 0, 2.0, 30)
    synth_case("n35_off3", 35, 200, 129, 3, 6.0, 40)
    chignolin_ref()
    print("done")
