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


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE config 3: ONE batch of 8192 frames (recipe S1), sharded 8 x 1024 across the GPUs of a node.  Round 5: the WHOLE batch through the
# real reference in fp32 and fp64 (tests/golden/full8192.npz, tools/make_goldens.py --only full8192: 2 x 12 minutes on 6 cores), so "is this
# a frame without an implementation-independent answer?" is answered by the reference's own pair on every frame -- the repo's C-oracle pair
# no longer fills in anywhere -- and the round-4 escape ("any distance, if the final loss is not above the oracle's") is gone.
# The tolerance enforced is the one BASELINE.md section 3 now states for S1 / S2:
#   * >= 99.9 % of the frames within 0.05 mm of the reference's fp32 run (max over the 22 joints), NO frame beyond 5 mm;
#   * every frame beyond 0.05 mm is flagged by the reference's own fp32 / fp64 pair (> 0.02 mm apart) or shows one of the two mechanisms of
#     tests/sensitivity.py (a LeakyReLU pre-activation within rounding of zero on its trajectory; a gradient component within rounding of
#     zero under Adam's first steps);
#   * there are at most 2 x (the pair's own count) + 2 of them.
@pytest.mark.parametrize("how", ["one_launch", "one_launch_w16", "shards_of_1024"])
def test_config3_whole_batch_against_the_reference(opts, dev, golden_dir, how):
    """one_launch: the 8192 frames on ONE GPU as DP_KERNEL_AUTO runs them (two rounds of dp_w4 since round 5); one_launch_w16: through the large-batch
    kernel (what AUTO took for this size through round 4, and takes beyond it).  shards_of_1024: the eight contiguous shards a
    node's eight ranks get (dragposer_amd.sharding.shard_bounds), each in its own launch with the kernel sharding.pick_kernel pins for the
    batch (dp_w4 at 1024 frames per GPU) -- every rank's slice held to the reference's rows of that slice, which is what
    tests/test_multi_gpu.py holds a real rank to when a node is there."""
    from dragposer_amd.optimizer import to_device_batch
    from dragposer_amd.sharding import pick_kernel, shard_bounds

    ref = R.load_golden(os.path.join(golden_dir, "full8192.npz"))
    mt = ref["meta"]
    B = mt["B"]
    assert B == 8192 and mt["n_iter"] == 50
    b = R.synth_inputs(R.OracleModel(), B, seed=mt["seed"])
    assert np.abs(b["tgt_pos"][:, T6] - ref["tgt_pos6"]).max() < 1e-5
    b["tgt_pos"][:, T6], b["tgt_rot"][:, T6] = ref["tgt_pos6"], ref["tgt_rot6"]
    assert _digest(b) == mt["digest"], "the recipe's inputs are not the ones the reference was run on"
    opt = opts["none"]
    if how.startswith("one_launch"):
        kern = "w16" if how.endswith("w16") else "auto"
        o = {k: v.cpu().numpy() for k, v in opt.optimize(**to_device_batch(b, dev), n_iter=50, lambda_tmp=mt["lambda_tmp"], kernel=kern).items()}
        assert opt.kernel_geometry()[0] == (64 if kern == "w16" else 16)  # dp_w16: one wave of 16 frames per SIMD; dp_w4: 4 waves of 4 frames
    else:
        kern = pick_kernel(opt, B, 8)
        assert kern == "w4"
        parts = []
        for r in range(8):
            lo, hi = shard_bounds(B, 8, r)
            parts.append({k: v.cpu().numpy() for k, v in
                          opt.optimize(**to_device_batch({k: b[k][lo:hi] for k in KEYS}, dev), n_iter=50, lambda_tmp=mt["lambda_tmp"], kernel=kern).items()})
        o = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    e = _mm(o["pos"], ref["pos"])
    err = e.max(axis=1)
    flag = np.zeros(B, bool)
    flag[ref["sens_frames"]] = True
    # on a frame the reference's own pair flags, its fp64 run is as good an answer as its fp32 run
    e64 = np.full(B, np.inf)
    e64[ref["sens_frames"]] = _mm(o["pos"][ref["sens_frames"]], ref["pos_f64_sens"]).max(axis=1)
    err = np.minimum(err, e64)
    bad = np.nonzero(err > 0.05)[0]
    ok, kink, tiny = explained(b, bad, 50, mt["lambda_tmp"], flagged=ref["sens_frames"])
    print(f"config 3, {how}: 8192 frames vs the reference's fp32 run: mean {e.mean():.5f} mm, p99 {np.percentile(e, 99):.5f}, max {err.max():.3f}; above 0.05 mm: "
          f"{bad.tolist()} ({np.round(err[bad], 3).tolist()} mm; smallest |pre-activation| {kink.tolist()}, smallest |dL/dz_k| {tiny.tolist()}); the reference's "
          f"own pair parts ways on {ref['sens_frames'].tolist()} ({np.round(ref['ref32_vs_ref64_mm'][ref['sens_frames']], 3).tolist()} mm)")
    assert (err <= 0.05).mean() >= 0.999 and err.max() <= 5.0, ((err > 0.05).sum(), err.max())
    assert ok.all() and len(bad) <= 2 * len(ref["sens_frames"]) + 2, (bad, kink, tiny)
    good = err <= 0.05
    assert e[good].mean() <= 0.002 and np.percentile(e[good], 99) <= 0.005
    # (loss terms of 1e-4 ... 1e-2: 2e-3 relative, or 1e-6 absolute on a term that is small against its frame's total)
    np.testing.assert_allclose(o["loss"][good], ref["loss_last"][good], rtol=2e-3, atol=1e-6)
    assert (o["iters"] == 50).all() and (o["status"] == 0).all()
    if how == "shards_of_1024":  # per rank: the bar holds on every slice, not just on the whole
        for r in range(8):
            lo, hi = shard_bounds(B, 8, r)
            assert (err[lo:hi] <= 0.05).mean() >= 0.997, (r, (err[lo:hi] > 0.05).sum())
