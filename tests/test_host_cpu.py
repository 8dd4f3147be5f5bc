"""Host-side logic that needs no GPU: C-ABI export check, weight flattening / checkpoint key
handling, CLI parsing, PDB writer, batch grouping and the multi-rank sharding (gloo, world 2)."""
import ctypes
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

import dff_amd
from dff_amd import binding, cli, pdbio, sampling, specs, weights
from oracle import synth

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    """Every function include/dff.h declares is exported by libdff_amd.so (no compute calls)."""
    header = open(os.path.join(ROOT, "include", "dff.h")).read()
    declared = set(re.findall(r"\b(dff_[a-z_0-9]+)\s*\(", header))
    declared -= {"dff_model", "dff_config", "dff_langevin_params"}
    assert declared == set(binding.SYMBOLS), (declared ^ set(binding.SYMBOLS))
    lib = binding.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dff_version()


def test_weight_count_and_bad_configs_without_gpu():
    lib = binding.load_library()
    for cfg, (_, N, H, L) in synth.SHIPPED_CONFIGS.items():
        c = binding.DffConfig(N, H, L, 1000, 1, 0, 0, 1)
        assert lib.dff_weight_count(ctypes.byref(c)) == synth.count_params(N, H, L)
    # the other input branches change the embedding shapes: node (H, N+1+3 abs), edge (H, 3 intr + dist)
    for intr, dist, ab in [(0, 1, 1), (1, 1, 1), (1, 0, 1), (1, 1, 0), (0, 1, 0), (0, 0, 0)]:
        c = binding.DffConfig(10, 64, 3, 1000, intr, dist, ab, 1)
        want = sum(int(np.prod(sh)) for _, sh, _, _ in synth.param_specs(10, 64, 3, 1, 11 + 3 * ab, (3 * intr + dist) or 1))
        assert lib.dff_weight_count(ctypes.byref(c)) == want
    # flag values other than 0 / 1 and a wrong weight count are rejected before any device work
    w = np.zeros(synth.count_params(10, 64, 3), np.float32)
    h = ctypes.c_void_p()
    c = binding.DffConfig(10, 64, 3, 1000, 1, 2, 0, 1)
    rc = lib.dff_model_create(ctypes.byref(c), w.ctypes.data_as(ctypes.c_void_p), w.size, 0, ctypes.byref(h))
    assert rc == 1 and b"must be 0 or 1" in lib.dff_last_error()
    c = binding.DffConfig(10, 64, 3, 1000, 1, 1, 0, 1)
    rc = lib.dff_model_create(ctypes.byref(c), w.ctypes.data_as(ctypes.c_void_p), w.size, 0, ctypes.byref(h))
    assert rc == 1 and b"expected" in lib.dff_last_error()
    c = binding.DffConfig(10, 80, 3, 1000, 1, 0, 0, 1)
    rc = lib.dff_model_create(ctypes.byref(c), w.ctypes.data_as(ctypes.c_void_p), w.size, 0, ctypes.byref(h))
    assert rc == 1 and b"hidden" in lib.dff_last_error()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(binding.DffLibraryError):
        binding.load_library(str(tmp_path / "nope.so"))


def test_flatten_order_and_shape_checks():
    _, N, H, L = synth.SHIPPED_CONFIGS["chignolin"]
    params = synth.synth_gnn_params(N, H, L)
    flat = weights.flatten_gnn_params(params, N, H, L)
    assert flat.dtype == np.float32 and flat.size == synth.count_params(N, H, L)
    # the ABI order is the reference's registration order == synth.param_specs order
    assert [k for k, _, _, _ in synth.param_specs(N, H, L)] == weights.abi_key_order(L)
    off = 0
    for k in weights.abi_key_order(L):
        n = params[k].size
        np.testing.assert_array_equal(flat[off:off + n], params[k].reshape(-1))
        off += n
    bad = dict(params)
    bad["node_decoder.weight"] = np.zeros((2, H), np.float32)
    with pytest.raises(ValueError):
        weights.flatten_gnn_params(bad, N, H, L)
    del bad["node_decoder.weight"]
    with pytest.raises(KeyError):
        weights.flatten_gnn_params(bad, N, H, L)


def test_checkpoint_layouts():
    _, N, H, L = synth.SHIPPED_CONFIGS["ala2"]
    params = {k: torch.from_numpy(v) for k, v in synth.synth_gnn_params(N, H, L).items()}
    gd = {"betas": torch.zeros(1000)}
    gd.update({"model." + k: v for k, v in params.items()})
    ema = {"initted": torch.tensor(True), "step": torch.tensor(5)}
    ema.update({"ema_model." + k: v for k, v in gd.items()})
    ema.update({"online_model." + k: v * 0 for k, v in gd.items()})  # must NOT be picked
    for data in ({"ema": ema, "model": gd, "step": 3}, {"model": gd}, gd, params):
        got = weights.gnn_params_from_checkpoint(data)
        assert set(got) == set(params)
        assert torch.equal(got["node_decoder.weight"], params["node_decoder.weight"])
    with pytest.raises(KeyError):
        weights.gnn_params_from_checkpoint({"ema": {"foo": 1}})


def test_cli_flags_match_reference():
    p = cli.build_parser()
    a = p.parse_args(["--model_path", "m"])
    assert (a.model_checkpoint, a.gen_mode, a.num_samples_eval, a.batch_size_gen) == ("best", "iid", 1000, 256)
    assert (a.friction, a.parallel_sim, a.n_timesteps, a.save_interval, a.noise_level) == (1, 100, 10000, 250, 20)
    assert a.dt is None and a.temp_data is None and a.temp_sim is None and a.kb == "consistent" and a.masses is None
    a = p.parse_args(["--model_path", "m", "--masses", "[12.0]*10", "--gen_mode", "langevin"])
    assert a.masses == [12.0] * 10
    assert p.parse_args(["--model_path", "m", "--masses", "[1, 2.5] + [3]*2"]).masses == [1.0, 2.5, 3.0, 3.0]
    with pytest.raises(SystemExit):
        p.parse_args(["--model_path", "m", "--masses", "__import__('os').system('true')"])
    with pytest.raises(SystemExit):
        p.parse_args([])  # --model_path required


def test_specs_tables():
    # bead counts: datasets/folded_pdbs CA records; std: dataset_utils_empty.py:38-48
    assert [specs.lookup(m).n_beads for m in ("CHIGNOLIN", "TRP_CAGE", "BBA", "VILLIN", "PROTEIN_G")] == [10, 20, 28, 35, 56]
    assert specs.lookup("alanine_dipeptide_fuberlin").n_beads == 5
    assert specs.norm_std("CHIGNOLIN") == 3.113133430480957
    assert specs.norm_std("alanine_dipeptide_fuberlin", fold=3) == 0.9452606439590454
    assert specs.default_masses("alanine_dipeptide_fuberlin") == [12.8] * 5
    assert specs.default_masses("VILLIN") == [12.0] * 35
    with pytest.raises(NotImplementedError):
        specs.lookup("unknown")


def test_pdb_writer(tmp_path):
    xyz = np.arange(2 * 10 * 3, dtype=np.float32).reshape(2, 10, 3) / 7.0
    out = tmp_path / "s.pdb"
    pdbio.save_pdb(str(out), xyz, "CHIGNOLIN")
    txt = out.read_text().splitlines()
    atoms = [l for l in txt if l.startswith("ATOM")]
    assert len(atoms) == 20 and sum(l.startswith("MODEL") for l in txt) == 2 and txt[-1] == "END"
    first = atoms[0]
    assert first[12:16].strip() == "CA" and first[17:20] == "TYR"
    assert float(first[30:38]) == pytest.approx(0.0) and float(atoms[1][30:38]) == pytest.approx(3 / 7, abs=1e-3)
    with pytest.raises(ValueError):
        pdbio.save_pdb(str(out), xyz[:, :5], "CHIGNOLIN")


def test_grouping_and_sharding():
    assert sampling.num_to_groups(1000, 256) == [256, 256, 256, 232]
    for total, world in [(256, 8), (100000, 8), (7, 3), (3, 8)]:
        spans = [sampling.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


_GLOO_WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    import dff_amd
    from dff_amd import sampling
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 7
    lo, hi = sampling.shard_range(total, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1).repeat(1, 2, 3)  # unit i holds value i
    full = sampling.gather_variable(local, total, world)
    assert full.shape == (total, 2, 3), full.shape
    assert torch.equal(full[:, 0, 0], torch.arange(total, dtype=torch.float32)), full[:, 0, 0]
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank%d-ok;" % rank); sys.stdout.flush()
""")


def test_two_rank_gather_gloo(tmp_path):
    """The N>1 path on CPU: shard_range + the single all_gather, world_size 2, gloo backend."""
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("-ok") == 2 and "rank0" in r.stdout and "rank1" in r.stdout, r.stdout


def test_bench_finds_the_profiled_traffic_for_the_shipped_kernels():
    """bench.py's roofline.traffic comes from the committed rocprofv3 PMC summary (profiles/<round>/traffic.json);
    rocprofv3 prints the full template argument list, the library the short kernel name: both must match."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.CHECK_PROFILE_SHA = False   # name / workload matching against every round's profiles; the hash check is the next test
    tr = lambda *a: bench.profile_figures(*a)["traffic"]   # noqa: E731
    t = tr("dff_small_kernel<64,8>", "chignolin", 256, 250)
    assert t is not None and 1e9 < t < 1e12
    assert tr("dff_fused_kernel<128,3,1,false>", "villin", 256, 250) is not None
    assert tr("dff_small_kernel<64,8>", "chignolin", 128, 250) is None   # other workload
    # round 2: the split-bf16 headline kernel (rocprofv3 prints "<64, 8, false, true>") has its own, smaller, traffic figure
    t2 = tr("dff_small_kernel<64,8,split_bf16>", "chignolin", 256, 250)
    assert t2 is not None and t2 < t
    # round 3: the headline kernel (rocprofv3: "<64, 8, false, true, true, 1>" -- the sampler mode is a template argument now)
    # keeps its activations in LDS / registers: its figure is the latest round's, two orders of magnitude below round 2's
    t3 = tr("dff_small_kernel<64,8,split_bf16,fold_kv>", "chignolin", 256, 250)
    assert t3 is not None and t3 < 0.1 * t2
    r = bench.roofline("chignolin", 256, 250, [20.8, 20.9], "dff_small_kernel<64,8,split_bf16>")
    assert abs(r["frac"] - 1.408e12 / 20.85e-3 / 157.3e12) < 1e-3 and r["traffic"] == t2
    # the engine's own roof: weight-GEMM FLOPs (2 L (8 N H I + 16 N H^2) = 19.66 of chignolin's 22.00
    # MFLOP) at the split products' rate, the rest at the fp32 rate, over the measured time
    assert abs(bench.weight_gemm_mflop("chignolin") - 2 * 3 * (8 * 10 * 64 * 512 + 16 * 10 * 64 * 64) / 1e6) < 1e-9
    ideal = lambda rate: (19.6608e6 / rate + (22.00e6 - 19.6608e6) / 157.3e12) * 256 * 250   # noqa: E731
    assert abs(r["mixed_peak_frac"] - ideal(2500e12 / 6) / 20.85e-3) < 2e-4
    r16 = bench.roofline("chignolin", 256, 250, [11.54] * 8, "dff_small_kernel<64,8,split_f16,fold_kv>")
    assert abs(r16["mixed_peak_frac"] - ideal(2500e12 / 3) / 11.54e-3) < 2e-4 and 0.20 < r16["mixed_peak_frac"] < 0.23
    r32 = bench.roofline("chignolin", 256, 250, [20.0] * 8, "dff_small_kernel<64,8>")
    assert abs(r32["mixed_peak_frac"] - r32["frac"]) < 2e-4          # the fp32 engine: one roof
    # round 4: the rocprof-reported MFMA utilisation and HBM rate ride in the roofline object (north_star), read from the
    # same profile directory as the traffic
    h = bench.roofline("chignolin", 256, 250, [13.3] * 8, "dff_small_kernel<64,8,split_bf16,fold_kv>")
    assert 0.05 < h["mfma_busy"] < 1.0 and 0.0 < h["hbm_tbps"] < 8.0 and h["l2_hit"] > 0.5 and h["profile"].startswith("profiles/r")
    v = bench.roofline("villin", 256, 250, [131.0] * 8, "dff_fused_kernel<128,3,1,false,split_bf16>", brief=True)
    assert 0.05 < v["mfma_busy"] < 1.0 and v["hbm_tbps"] > 0.1 and "peak" not in v
    assert v["profile"].startswith("profiles/r")   # (the brief entries name their profile directory too)
    # ... and the line must fit the driver's stdout tail: every `also` entry <= 700 bytes, what they share said once
    import json
    entry = {"workload": "villin (35 beads, H=128, L=3) Langevin, 256/GPU", "value": 1897.33, "unit": "MD-steps/s (batch 256, whole job)",
             "dtype": bench.kernel_dtype("dff_fused_kernel<128,3,1,false,split_f16>"), "ms_per_step": 0.52712, "steps": 2000, "finite": True, "roofline": v,
             "cpu_baseline": {"value": 0.3312, "cores": 16, "kind": "port", "sample": "3 Langevin steps, same workload (P=256, villin), 16 of 256 logical cores"}}
    assert len(json.dumps(entry)) <= 700 and len(json.dumps(bench.NOTES)) <= 1400
    # `dtype` says which engine multiplied the weights (derived from the kernel name)
    assert bench.kernel_dtype("dff_small_kernel<64,8>") == "f32" and "3-way bf16 split" in bench.NOTES["dtype"]
    assert "2xf16" in bench.kernel_dtype("dff_small_kernel<64,8,split_f16,fold_kv>") and bench.kernel_dtype("x<split_f16>").startswith("f32")
    assert "3xbf16" in bench.kernel_dtype("dff_fused_kernel<128,3,1,false,split_bf16>")
    # algorithmic FLOPs per launch of the headline config: 22.00 MFLOP x 256 x 250
    assert abs(bench.MFLOP_PER_CALL["chignolin"] * 1e6 * 256 * 250 - 1.408e12) < 1e6


def test_bench_drops_profile_counters_taken_on_other_sources(tmp_path):
    """a traffic.json is tied to the sources it profiled (src_sha = sha256 over csrc/* + include/dff.h,
    compiled into the library by build.sh, stamped by tools_profile_report.py).  bench.py reports the rocprofv3 counters
    only next to a library built from the same sources; otherwise they are None and roofline.profile_stale is true."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from dff_amd import srcsha
    from dff_amd.binding import load_library
    # the library carries the hash of the tree it was built from (build.sh) -- and the tree has not changed since
    lib_sha = srcsha.library_sha(load_library())
    assert len(lib_sha) == 16 and lib_sha == srcsha.tree_sha(), "libdff_amd.so is older than its sources: run ./build.sh"
    assert bench.library_src_sha() == lib_sha
    d = tmp_path / "profiles" / "r99" / "x"
    d.mkdir(parents=True)
    t = {"workload": "chignolin P=256 chunk=250", "kernel": "dff_small_kernel<64, 8, false, true, true, 1>",
         "hbm_bytes_per_launch": 1.3e8, "avg_launch_ms": 13.2, "hbm_tbps": 0.0098, "mfma_busy": 0.33, "l2_hit": 0.999,
         "src_sha": lib_sha}
    json.dump(t, open(d / "traffic.json", "w"))
    bench.ROOT = str(tmp_path)
    kn = "dff_small_kernel<64,8,split_bf16,fold_kv>"
    ok = bench.roofline("chignolin", 256, 250, [13.2] * 8, kn)
    assert ok["traffic"] == 1.3e8 and ok["mfma_busy"] == 0.33 and "profile_stale" not in ok and ok["profile"] == "profiles/r99/x"
    t["src_sha"] = "0123456789abcdef"   # the kernel changed after this profile was taken
    json.dump(t, open(d / "traffic.json", "w"))
    stale = bench.roofline("chignolin", 256, 250, [13.2] * 8, kn)
    assert stale["profile_stale"] is True and stale["traffic"] is None and stale["mfma_busy"] is None and stale["hbm_tbps"] is None
    assert stale["frac"] == ok["frac"]   # the timing figures do not depend on the profile
    del t["src_sha"]                     # a profile from before round 5 (no hash): stale as well
    json.dump(t, open(d / "traffic.json", "w"))
    assert bench.roofline("chignolin", 256, 250, [13.2] * 8, kn, brief=True)["profile_stale"] is True
