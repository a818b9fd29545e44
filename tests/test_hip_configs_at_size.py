"""GPU (MI355X): BASELINE configs 4 and 5 AT SIZE -- 1024 frames each -- through both kernels, against the REAL reference's run of
the same inputs in fp32 and in fp64 (tests/golden/full_s3_1024.npz, full_s4_1024.npz: tools/make_goldens.py --only
full_s3_1024,full_s4_1024; DragPose.run frame by frame), in BASELINE's own terms:

  config 4 / recipe S3 (3 trackers [13,17,21], 100 iterations, lambda_temporal 0.15):   mean <= 0.05 mm, p99 <= 1 mm, max <= 3 mm
  config 5 / recipe S4 (1-6 trackers per frame, bf16-rounded decoder weight tensors):    max <= 0.5 mm (10 x S1's 0.05)

over all B x 22 joint positions.  The only allowance: frames without an implementation-independent answer -- those on which the
reference's OWN fp32 and fp64 runs part ways (> 0.02 mm), and those whose fp32 trajectory takes a LeakyReLU pre-activation within
rounding of zero (tests/sensitivity.py) -- are bounded at 5 mm and counted.  On top of BASELINE's bars: a kernel's p99 must
stay within 2 x the reference pair's own p99 (+ 0.001 mm of fp32 re-association noise), its count of frames above 0.05 mm within
2 x the pair's count of frames above 0.02 mm (at least 2), and every such frame must show one of the two mechanisms of
tests/sensitivity.py (a LeakyReLU pre-activation within rounding of zero; a gradient component within rounding of zero under Adam's
first steps) unless the reference pair flags it itself.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from sensitivity import explained  # tests/sensitivity.py

pytestmark = pytest.mark.gpu

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")
T6 = [0, 3, 7, 13, 17, 21]
CASES = {
    # name: recipe arguments of oracle.ref_torch.synth_inputs, weight rounding, BASELINE's (mean, p99, max) in mm
    "full_s3_1024": dict(trackers=3, mixed=False, wd="none", bar=(0.05, 1.0, 3.0)),
    "full_s4_1024": dict(trackers=6, mixed=True, wd="bf16", bar=(None, None, 0.5)),
}


def _mm(a, b):
    return np.linalg.norm(a - b, axis=-1) * 1000.0


def _digest(b):
    import hashlib

    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(b[k]).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def opts(dev):
    from dragposer_amd.optimizer import LatentOptimizer

    return {"none": LatentOptimizer(device=dev), "bf16": LatentOptimizer(device=dev, weight_dtype="bf16")}


def load_case(golden_dir, name):
    """the recipe's inputs as the reference saw them (targets as drawn on the generating host: CPU matrix products differ in their
    last bits from host to host, the random draws do not) and the reference's two runs"""
    c = CASES[name]
    ref = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    mt = ref["meta"]
    b = R.synth_inputs(R.OracleModel(weight_rounding=c["wd"]), mt["B"], trackers=c["trackers"], mixed=c["mixed"], seed=mt["seed"])
    trk6 = b["tracked"][:, T6].astype(bool)
    if c["mixed"]:
        np.testing.assert_array_equal(trk6, ref["tracked6"].astype(bool))
    assert np.abs(b["tgt_pos"][:, T6] - ref["tgt_pos6"]).max() < 1e-5
    b["tgt_pos"][:, T6], b["tgt_rot"][:, T6] = ref["tgt_pos6"], ref["tgt_rot6"]
    assert _digest(b) == mt["digest"], "the recipe's inputs are not the ones the reference was run on"
    return c, ref, mt, b


@pytest.mark.parametrize("kernel", ["w4", "w16"])
@pytest.mark.parametrize("name", list(CASES))
def test_config_at_size_against_the_reference(opts, dev, golden_dir, name, kernel):
    from dragposer_amd.optimizer import to_device_batch

    c, ref, mt, b = load_case(golden_dir, name)
    o = opts[c["wd"]].optimize(**to_device_batch(b, dev), n_iter=mt["n_iter"], lambda_tmp=mt["lambda_tmp"], kernel=kernel)
    torch.cuda.synchronize()
    o = {k: v.cpu().numpy() for k, v in o.items()}
    e = _mm(o["pos"], ref["pos"])            # [B, 22] against the reference's fp32 run
    pair = _mm(ref["pos"], ref["pos_f64"])   # the reference's own fp32 vs fp64 runs
    flagged = pair.max(axis=1) > 0.02
    err = e.max(axis=1)
    miss = np.nonzero(err > 0.05)[0]
    ok, kink, tiny = (explained(b, miss, mt["n_iter"], mt["lambda_tmp"], flagged=np.nonzero(flagged)[0], weight_rounding=c["wd"])
                      if len(miss) else (np.zeros(0, bool), [], []))
    e64 = _mm(o["pos"], ref["pos_f64"]).max(axis=1)
    print(f"{name} ({kernel}) vs the reference's fp32 run over B x 22: mean {e.mean():.5f} mm, p99 {np.percentile(e, 99):.5f}, max {e.max():.4f} "
          f"(BASELINE {c['bar']}); the reference's own fp32 vs fp64: mean {pair.mean():.5f}, p99 {np.percentile(pair, 99):.5f}, max {pair.max():.4f}, "
          f"frames above 0.02 mm {np.nonzero(flagged)[0].tolist()}; kernel above 0.05 mm: frames {miss.tolist()} ({np.round(err[miss], 3).tolist()} mm; "
          f"vs the fp64 run {np.round(e64[miss], 3).tolist()}; smallest |pre-activation| {list(kink)}, smallest |dL/dz_k| {list(tiny)})")
    mean_bar, p99_bar, max_bar = c["bar"]
    if mean_bar is not None:
        assert e.mean() <= mean_bar and np.percentile(e, 99) <= p99_bar, (e.mean(), np.percentile(e, 99))
    # BASELINE's max, off the frames without an implementation-independent answer: the ones the reference's own pair flags and the ones
    # that show the mechanism by which two correct fp32 implementations part ways (a LeakyReLU pre-activation within rounding of zero on
    # the path -- profiles/r04_divergence_s4.txt shows it for frame 962 of recipe S4 under dp_w16: unit 5 at 4e-7 / 1e-6 with opposite
    # signs for the kernel and the fp32 oracle at iteration 34); those are counted below and bounded here
    free = ~flagged
    free[miss[ok]] = False
    assert err[free].max() <= max_bar, (err[free].max(), np.nonzero(free & (err > max_bar))[0])
    assert err.max() <= 5.0
    # beyond BASELINE: as close to the reference's fp32 run as its own fp64 run is
    assert np.percentile(e, 99) <= 2.0 * np.percentile(pair, 99) + 0.001, (np.percentile(e, 99), np.percentile(pair, 99))
    assert len(miss) <= max(2, 2 * int(flagged.sum())), (miss, flagged.sum())
    # Every miss shows a mechanism.  The thresholds of tests/sensitivity.py were calibrated on 50-iteration, 6-tracker frames; recipe S3
    # runs 100 iterations with the legs constrained by nothing but the shared latent (flat valleys: a difference in the 6th digit
    # grows for 100 steps), so there a miss may instead simply be small -- within a third of BASELINE's max.
    assert all(k or (c["trackers"] == 3 and x <= 1.0) for k, x in zip(ok, err[miss])), (miss, err[miss], kink, tiny)
    good = err <= 0.05
    np.testing.assert_allclose(o["loss"][good], ref["loss_last"][good], rtol=2e-3, atol=1e-8)
    assert (np.abs(o["z_pre"] - ref["z_pre"])[good] <= 2e-4).mean() >= 0.999  # (flat latent directions: positions agree, a component may not)
    assert (o["iters"] == mt["n_iter"]).all()


@pytest.mark.parametrize("name", list(CASES))
def test_kernels_agree_with_each_other_at_size(opts, dev, golden_dir, name):
    """the two kernels frame by frame on the same 1024 frames: independent arithmetic (fp32 MFMA vs split-bf16 MFMA, other lane
    layout of the kinematics), so their distance is of the size of either one's distance from the reference"""
    from dragposer_amd.optimizer import to_device_batch

    c, ref, mt, b = load_case(golden_dir, name)
    d = to_device_batch(b, dev)
    a4 = opts[c["wd"]].optimize(**d, n_iter=mt["n_iter"], lambda_tmp=mt["lambda_tmp"], kernel="w4", outputs=("pos",))["pos"].cpu().numpy()
    a16 = opts[c["wd"]].optimize(**d, n_iter=mt["n_iter"], lambda_tmp=mt["lambda_tmp"], kernel="w16", outputs=("pos",))["pos"].cpu().numpy()
    e = _mm(a4, a16)
    print(f"{name}: w4 vs w16 mean {e.mean():.5f} mm, p99 {np.percentile(e, 99):.5f}, max {e.max():.4f}, frames above 0.05 mm: {np.nonzero(e.max(1) > 0.05)[0].tolist()}")
    assert np.percentile(e, 99) <= 0.01 and (e.max(axis=1) > 0.05).sum() <= 4
