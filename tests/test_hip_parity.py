"""GPU (MI355X): the HIP path, called through the C ABI, against the oracle and the golden vectors.

Tolerance (FK joint positions, all 22 joints, millimetres, evaluated at the latent of the last
forward pass, like the reference's returned pose): max <= 0.05 mm (observed fp32
re-association noise is ~0.001 mm) on every frame that is WELL-CONDITIONED.  The loss is continuous
but its gradient is not: the decoder's two LeakyReLU layers switch slope (1 <-> 0.2) where a hidden
pre-activation crosses zero.  On a frame whose trajectory takes some pre-activation within fp32
rounding of zero (|pre| of a few 1e-6 or less; typical minimum over 50 iterations: 3e-4), two correct
implementations pick different slopes for that unit, their gradients differ by O(1) on its path, and
the optimisation continues down another route.  WHICH frames do so depends on the last bit of every
implementation's arithmetic; THAT some do is a property of the data: the reference's own fp32 and fp64 runs
(same code, same inputs; stored in the fixtures by tools/make_goldens.py) part ways on 1 of the 64 frames of the
3-tracker fixture (2.1 mm) and on 1 of the 4096 frames of the headline batch (3.2 mm).  Frames the reference
flags that way, and frames on which `_kink_distance` shows the mechanism, are bounded at a few mm plus agreement
of the final loss; everything else is held to the strict 0.05 mm.  Where no reference run exists (inputs made up
in a test) the repo's fp32 / fp64 C oracles play the reference pair's role (`_sensitive_frames`).
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R
from oracle.analytic import AnalyticOracle
from sensitivity import kink_distance as _kink_distance, tiny_gradient as _tiny_gradient  # tests/sensitivity.py

pytestmark = pytest.mark.gpu

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")


def _mm(a, b):
    return np.linalg.norm(a - b, axis=-1) * 1000.0


def _sensitive_frames(b, n_iter, lam, weight_rounding="none"):
    """frames on which the oracle's own fp32 and fp64 runs part ways (> 0.02 mm at some joint)"""
    a = [b[k] for k in KEYS]
    o32 = AnalyticOracle(precision="f32", weight_rounding=weight_rounding).optimize(*a, n_iter, lam_tmp=lam)
    o64 = AnalyticOracle(precision="f64", weight_rounding=weight_rounding).optimize(*a, n_iter, lam_tmp=lam)
    return _mm(o32["pos"], o64["pos"]).max(axis=1) > 0.02, o32


def _ref_sensitive(g):
    """Frames on which the REFERENCE's own fp32 and fp64 runs (same code, same inputs; tools/make_goldens.py --only NAME_f64)
    part ways by more than 0.02 mm at some joint: sensitivity as a property of the data, stated without the repo's oracles."""
    return _mm(g["pos"], g["pos_f64"]).max(axis=1) > 0.02


def _inputs_digest(b):
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
def opt(dev):
    from dragposer_amd.optimizer import LatentOptimizer

    return LatentOptimizer(device=dev)


@pytest.fixture(scope="module")
def opt_bf16(dev):
    from dragposer_amd.optimizer import LatentOptimizer

    return LatentOptimizer(device=dev, weight_dtype="bf16")


def _run(o, g, n_iter, lam, **kw):
    from dragposer_amd.optimizer import to_device_batch

    out = o.optimize(**to_device_batch(g, o.device), n_iter=n_iter, lambda_tmp=lam, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def test_native_library_is_what_runs(opt):
    from dragposer_amd import _lib

    assert os.path.exists(_lib.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libdragposer_hip.so" in maps
    assert opt.frames_per_block == 16 and opt.threads_per_block == 256  # dp_w4.hip: four waves of four frames


def test_forward_matches_oracle(opt, dev):
    m = R.OracleModel()
    b = R.synth_inputs(m, 100)  # ragged: 100 = 6 blocks + 4 frames
    o = opt.forward(torch.from_numpy(b["z_src"]).to(dev), torch.from_numpy(b["cur_rot"]).to(dev))
    torch.cuda.synchronize()
    f = AnalyticOracle(precision="f64").forward(b["z_src"], b["cur_rot"])
    assert _mm(o["pos"].cpu().numpy(), f["pos"]).max() < 0.005
    np.testing.assert_allclose(o["rot"].cpu().numpy(), f["rot"], atol=1e-5)
    np.testing.assert_allclose(o["world_rot"].cpu().numpy(), f["world_rot"], atol=2e-6)
    np.testing.assert_allclose(o["world_disp"].cpu().numpy(), f["world_disp"], atol=2e-7)
    np.testing.assert_allclose(o["pose"].cpu().numpy(), f["pose"], atol=1e-3)  # normalised space (/sigma ~ 1700x)
    # the targets of recipe S are exactly this forward pass
    trk = b["tracked"].astype(bool)
    assert _mm(o["pos"].cpu().numpy()[trk], b["tgt_pos"][trk]).max() < 0.005


@pytest.mark.parametrize("name", ["s1", "s3", "s4"])
def test_first_iteration_gradient_and_losses(opt, opt_bf16, dev, golden_dir, name):
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    mt = g["meta"]
    o = opt_bf16 if mt["weight_rounding"] == "bf16" else opt
    dbg = torch.zeros(len(g["z0"]), 240, device=dev)
    out = o.optimize(**to_device_batch(g, dev), n_iter=1, lambda_tmp=mt["lambda_tmp"], _debug=dbg)
    torch.cuda.synchronize()
    A = AnalyticOracle(precision="f64", weight_rounding=mt["weight_rounding"])
    lo, gr = A.grad(*[g[k] for k in KEYS], 1.0, mt["lambda_tmp"])
    np.testing.assert_allclose(dbg.cpu().numpy()[:, 208:232], gr, atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(out["loss"].cpu().numpy(), g["loss_hist"][:, 0], rtol=1e-5, atol=1e-9)  # reference's own losses
    # Adam step 1 moves every component by exactly lr
    dz = out["z"].cpu().numpy() - g["z0"]
    big = np.abs(gr) > 1e-3  # where eps=1e-8 is negligible against |g|
    np.testing.assert_allclose(dz[big], -0.01 * np.sign(gr[big]), atol=2e-7)
    ref1 = A.optimize(*[g[k] for k in KEYS], 1, lam_tmp=mt["lambda_tmp"])["z_final"]
    np.testing.assert_allclose(out["z"].cpu().numpy(), ref1, atol=1e-5)


@pytest.mark.parametrize("name", ["s1", "s4"])
def test_golden_parity_6_trackers(opt, opt_bf16, golden_dir, name):
    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    mt = g["meta"]
    o = _run(opt_bf16 if mt["weight_rounding"] == "bf16" else opt, g, mt["n_iter"], mt["lambda_tmp"])
    err = _mm(o["pos"], g["pos"])
    assert not _ref_sensitive(g).any()  # (the reference's fp32 and fp64 runs agree on every frame of these fixtures)
    print(f"{name}: vs the reference's fp32 run max {err.max():.5f} mm, vs its fp64 run {_mm(o['pos'], g['pos_f64']).max():.5f} mm; "
          f"the reference's own pair {_mm(g['pos'], g['pos_f64']).max():.5f} mm")
    assert err.max() <= (0.5 if mt["weight_rounding"] == "bf16" else 0.05), err.max()  # S4 = 10 x S1
    np.testing.assert_allclose(o["z"], g["z_final"], atol=5e-5)
    np.testing.assert_allclose(o["z_pre"], g["z_pre"], atol=5e-5)
    np.testing.assert_allclose(o["world_rot"], g["world_rot"], atol=5e-6)
    np.testing.assert_allclose(o["world_disp"], g["world_disp"], atol=5e-7)
    np.testing.assert_allclose(o["rot"], g["rot"], atol=2e-5)
    np.testing.assert_allclose(o["pose"], g["pose"], atol=2e-3)
    np.testing.assert_allclose(o["loss"], g["loss_hist"][:, -1], rtol=2e-3, atol=1e-8)
    assert np.all(o["iters"] == mt["n_iter"])


def test_golden_parity_3_trackers_100_iters(opt, golden_dir):
    """BASELINE config 4 / SURVEY 8d recipe S3 (3 trackers, 100 iterations, under-constrained legs) against the reference's own
    run, in BASELINE's terms: over all B x 22 joint positions mean <= 0.05 mm and p99 <= 1 mm; max <= 3 mm -- except on frames
    the REFERENCE itself cannot reproduce between fp32 and fp64 (frame 14 of this fixture: 2.14 mm between its two runs), where
    an implementation-independent answer does not exist and the bound is 10 mm plus an optimum of the same quality."""
    g = R.load_golden(os.path.join(golden_dir, "s3.npz"))
    mt = g["meta"]
    o = _run(opt, g, mt["n_iter"], mt["lambda_tmp"])
    e = _mm(o["pos"], g["pos"])  # [B, 22]
    err = e.max(axis=1)
    sens = _ref_sensitive(g)
    e64 = _mm(o["pos"], g["pos_f64"]).max(axis=1)
    ref_pair = _mm(g["pos"], g["pos_f64"])
    print(f"s3 vs the reference's fp32 run: mean {e.mean():.4f} mm, p99 {np.percentile(e, 99):.4f} mm, max {e.max():.4f} mm (BASELINE: 0.05 / 1 / 3); "
          f"the reference's own fp32 vs fp64: mean {ref_pair.mean():.4f}, p99 {np.percentile(ref_pair, 99):.4f}, max {ref_pair.max():.4f}; "
          f"reference-flagged frames {np.nonzero(sens)[0].tolist()}: GPU vs ref fp32 {np.round(err[sens], 3).tolist()} mm, vs ref fp64 "
          f"{np.round(e64[sens], 3).tolist()} mm; largest error elsewhere {err[~sens].max():.4f} mm")
    assert 0 < sens.sum() <= 3
    assert e.mean() <= 0.05 and np.percentile(e, 99) <= 1.0, (e.mean(), np.percentile(e, 99))
    assert err[~sens].max() <= 0.05, err[~sens].max()  # (BASELINE allows 3 mm here; observed 0.011)
    assert err[sens].max() <= 10.0, err[sens].max()
    np.testing.assert_allclose(o["z"][~sens], g["z_final"][~sens], atol=5e-4)
    tot_ref, tot = g["loss_hist"][:, -1].sum(1), o["loss"].sum(1)
    np.testing.assert_allclose(tot[~sens], tot_ref[~sens], rtol=2e-3)
    np.testing.assert_allclose(tot[sens], tot_ref[sens], rtol=0.1)  # same optimum quality where the path diverged


def test_early_stop_matches_reference_loop(opt, golden_dir):
    """The reference's own eval settings (eval_drag.py:210-214): stop_eps_pos 1e-4, stop_eps_rot 1e-2,
    max_iter 100, min_loss_incr 1e-5 -- iteration counts, returned pose and stepped latent per frame."""
    g = R.load_golden(os.path.join(golden_dir, "es.npz"))
    mt = g["meta"]
    o = _run(opt, g, mt["n_iter"], mt["lambda_tmp"], stop_eps_pos=mt["stop_eps_pos"], stop_eps_rot=mt["stop_eps_rot"],
             min_loss_incr=mt["min_loss_incr"])
    assert g["iters"].min() < 20 and g["iters"].max() == 100  # the fixture exercises both exits
    same = o["iters"] == g["iters"]
    # the loss increment sits within rounding of the 1e-5 threshold on a few frames: allow +-1 iteration on <= 5 %
    assert same.mean() >= 0.95 and np.abs(o["iters"] - g["iters"]).max() <= 1, (same.mean(), np.abs(o["iters"] - g["iters"]).max())
    assert _mm(o["pos"][same], g["pos"][same]).max() <= 0.05
    np.testing.assert_allclose(o["z"][same], g["z_final"][same], atol=5e-5)
    np.testing.assert_allclose(o["z_pre"][same], g["z_pre"][same], atol=5e-5)
    last = g["loss_hist"][np.arange(len(g["iters"])), g["iters"] - 1]
    np.testing.assert_allclose(o["loss"][same], last[same], rtol=2e-3, atol=1e-8)


@pytest.mark.parametrize("B", [1, 15, 17, 33])
def test_ragged_batches_equal_full_batch_rows(opt, golden_dir, B):
    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    full = _run(opt, g, 20, 0.02)
    sub = _run(opt, {k: g[k][:B] for k in KEYS}, 20, 0.02)
    for k in ("z", "pos", "pose", "loss"):
        np.testing.assert_array_equal(sub[k], full[k][:B])  # a frame's result never depends on its batch


def test_launch_plan_repeats_the_call_over_the_same_buffers(opt, golden_dir):
    """LatentOptimizer.plan: arguments marshalled once, then one dp_optimize call per `plan()`.  Same bits as `optimize`; the inputs are
    read at launch time, so refilling them in place moves the result; a plan over bad arguments is refused when it is made."""
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    want = _run(opt, g, 20, 0.02)
    b = to_device_batch(g, opt.device)
    plan = opt.plan(**b, n_iter=20, lambda_tmp=0.02)
    out = plan()
    assert out is plan.results and set(out) == set(want)
    torch.cuda.synchronize()
    for k in ("z", "pos", "pose", "loss", "iters", "status"):
        np.testing.assert_array_equal(out[k].cpu().numpy(), want[k])
    first = out["z"].clone()
    b["z0"].copy_(torch.roll(b["z0"], 1, 0))  # in place: the plan holds the pointer, not a copy
    plan()
    torch.cuda.synchronize()
    g2 = {k: g[k] for k in KEYS}
    g2["z0"] = np.roll(g["z0"], 1, 0)
    np.testing.assert_array_equal(out["z"].cpu().numpy(), _run(opt, g2, 20, 0.02)["z"])
    assert not torch.equal(first, out["z"])
    with pytest.raises(ValueError):
        opt.plan(**{**b, "z0": b["z0"][:, :23]}, n_iter=20)
    # a plan outliving its optimiser is refused, not run on a freed context
    from dragposer_amd._lib import DragPoserError
    from dragposer_amd.optimizer import LatentOptimizer

    o2 = LatentOptimizer(device=opt.device)
    p2 = o2.plan(**b, n_iter=2)
    p2()
    torch.cuda.synchronize()
    o2.close()
    with pytest.raises(DragPoserError):
        p2()


@pytest.mark.parametrize("kernel", ["auto", "w16"])
def test_full_size_batch_properties(opt, dev, golden_dir, kernel):
    """BASELINE's headline batch (4096 frames x 50 iterations): determinism, batch-position invariance, and parity on EVERY
    frame against the REAL reference's fp32 run of the same inputs (tests/golden/full4096.npz: the reference's DragPose.run,
    frame by frame, in fp32 and in fp64).  "auto" is dp_w4 at this size; "w16" holds the large-batch kernel (split-bf16
    products, what "auto" launches beyond 4096 frames) to the same reference run and the same bar."""
    from dragposer_amd.optimizer import to_device_batch

    m = R.OracleModel()
    b = R.synth_inputs(m, 4096)
    ref = R.load_golden(os.path.join(golden_dir, "full4096.npz"))
    # the targets as drawn on the host that ran the reference (they come out of CPU matrix products, whose last bits differ from
    # host to host; the random draws themselves do not): with them the inputs are exactly the ones the reference saw
    T6 = [0, 3, 7, 13, 17, 21]
    assert np.abs(b["tgt_pos"][:, T6] - ref["tgt_pos6"]).max() < 1e-5
    b["tgt_pos"][:, T6], b["tgt_rot"][:, T6] = ref["tgt_pos6"], ref["tgt_rot6"]
    assert _inputs_digest(b) == ref["meta"]["digest"], "the recipe's inputs are not the ones the reference was run on"
    d = to_device_batch(b, dev)
    o1 = {k: v.cpu().numpy() for k, v in opt.optimize(**d, n_iter=50, kernel=kernel).items()}
    o2 = {k: v.cpu().numpy() for k, v in opt.optimize(**d, n_iter=50, kernel=kernel).items()}
    for k in o1:
        np.testing.assert_array_equal(o1[k], o2[k])  # bitwise reproducible
    perm = np.random.RandomState(0).permutation(4096)
    dp = to_device_batch({k: b[k][perm] for k in KEYS}, dev)
    o3 = opt.optimize(**dp, n_iter=50, kernel=kernel)
    np.testing.assert_array_equal(o3["z"].cpu().numpy(), o1["z"][perm])
    # Every frame against the reference.  The loss's gradient is discontinuous where a LeakyReLU pre-activation crosses zero
    # (module docstring): on a frame whose trajectory takes one within fp32 rounding of zero, two correct implementations part
    # ways -- the reference's own fp32 and fp64 runs do on 1 of these 4096 frames (frame 2327, 3.2 mm), each other pair of
    # implementations on its own one or two.  So: at most 4 frames (0.1 %) above 0.05 mm, none above 5 mm, and every one of them
    # either reference-flagged or showing a mechanism: |pre-activation| < 5e-6 on its fp64 trajectory (typical frames 3e-4), or
    # -- rarer, small -- a component of dL/dz below 1e-5 in the first iterations (`_tiny_gradient`).  Where a miss starts and what
    # sits there, for the 20 missed frames of 12 more seeds: profiles/r03_soak_divergence.txt (tools/soak_divergence.py).
    e = _mm(o1["pos"], ref["pos"])
    err = e.max(axis=1)
    ref_flag = np.zeros(4096, bool)
    ref_flag[ref["sens_frames"]] = True
    allowance = np.nonzero(err > 0.05)[0]
    kink = _kink_distance(b, allowance, 50, 0.02)
    tiny = _tiny_gradient(b, allowance, 0.02)
    print(f"{kernel}: 4096 frames vs the reference's fp32 run: mean {e.mean():.5f} mm, p99 {np.percentile(e, 99):.5f}, p99.9 {np.percentile(e, 99.9):.5f}, max {e.max():.3f}; "
          f"above 0.05 mm: frames {allowance.tolist()} ({np.round(err[allowance], 3).tolist()} mm, smallest |pre-activation| {kink.tolist()}); "
          f"the reference's own fp32 vs fp64 runs: frames {ref['sens_frames'].tolist()} ({np.round(ref['ref32_vs_ref64_mm'][ref['sens_frames']], 3).tolist()} mm)")
    assert len(allowance) <= 4 and err.max() <= 5.0, (allowance, err.max())
    assert all(ref_flag[f] or k < 5e-6 or t < 1e-5 for f, k, t in zip(allowance, kink, tiny)), (allowance, kink, tiny)
    assert np.percentile(err, 99.8) <= 0.05 and err[err <= 0.05].mean() <= 0.002, (np.percentile(err, 99.8), err.mean())
    dz = np.abs(o1["z"] - ref["z_final"])[err <= 0.05]  # latent: 5e-5 on all but one (dp_w4) / 13 (dp_w16) of 98 000 components -- flat
    assert (dz <= 5e-5).mean() >= 0.9998 and dz.max() <= 1e-3, ((dz > 5e-5).sum(), dz.max())  # latent directions, positions within 0.05 mm
    np.testing.assert_allclose(o1["loss"][err <= 0.05], ref["loss_last"][err <= 0.05], rtol=2e-3, atol=1e-8)
    first = opt.optimize(**d, n_iter=1, kernel=kernel)["loss"].cpu().numpy().sum(1)
    assert (o1["loss"].sum(1) < first).mean() > 0.99
    assert np.isfinite(o1["z"]).all()


@pytest.mark.parametrize("n_trk", [9, 22])
def test_more_than_six_trackers_general_path(opt, dev, n_trk):
    """Any joint may carry a tracker (the kernel's first 6 ranks are the fast path): 9 and all 22 joints, with
    per-joint weights, against the C oracle."""
    from dragposer_amd.optimizer import to_device_batch

    m = R.OracleModel()
    b = R.synth_inputs(m, 48)
    rs = np.random.RandomState(3)
    with torch.no_grad():
        mo, d = R.decoder_forward(m, torch.tensor(b["z_src"]))
        _, _, pos, rot, _ = R.pose_fk(m, mo, d, torch.tensor(b["cur_rot"]))
    b["tracked"][:] = 0
    b["w"][:] = 0
    for f in range(48):
        js = np.sort(rs.permutation(22)[:n_trk])
        b["tracked"][f, js] = 1
        b["w"][f, js, 0] = rs.uniform(1, 10, n_trk)
        b["w"][f, js, 1] = rs.uniform(0.01, 2, n_trk)
    trk = b["tracked"].astype(bool)[..., None]
    b["tgt_pos"] = (pos.numpy() * trk).astype(np.float32)
    b["tgt_rot"] = (rot.numpy().reshape(48, 22, 9) * trk).astype(np.float32)
    o = {k: v.cpu().numpy() for k, v in opt.optimize(**to_device_batch(b, dev), n_iter=30).items()}
    sens, ref = _sensitive_frames(b, 30, 0.02)
    err = _mm(o["pos"], ref["pos"]).max(axis=1)
    assert sens.sum() <= 2 and err[~sens].max() <= 0.05, (sens.sum(), err[~sens].max())
    np.testing.assert_allclose(o["loss"][~sens], ref["loss"][~sens], rtol=2e-3, atol=1e-8)


def test_argument_validation(opt, dev, golden_dir):
    from dragposer_amd import _lib
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    d = to_device_batch(g, dev)
    with pytest.raises(_lib.DragPoserError) as e:
        opt.optimize(**d, n_iter=_lib.DP_MAX_ITERS + 1)
    assert e.value.code == _lib.DP_ERR_INVALID
    with pytest.raises(ValueError):
        opt.optimize(**{**d, "w": d["w"].double()}, n_iter=5)
    with pytest.raises(ValueError):
        opt.optimize(**{**d, "z0": d["z0"].cpu()}, n_iter=5)


def test_edge_sizes(opt, dev, golden_dir):
    """empty batch and iteration-count limits are refused; the maximum iteration count and a quarter-million-frame
    batch run, and a frame's result does not depend on where in a large batch it sits"""
    from dragposer_amd import _lib
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    d = to_device_batch(g, dev)
    with pytest.raises(_lib.DragPoserError) as e:
        opt.optimize(**{k: v[:0] for k, v in d.items()}, n_iter=5)
    assert e.value.code == _lib.DP_ERR_INVALID
    with pytest.raises(_lib.DragPoserError):
        opt.optimize(**d, n_iter=0)
    o = opt.optimize(**{k: v[:16] for k, v in d.items()}, n_iter=256)  # the whole argument table of Adam scalars
    assert torch.isfinite(o["z"]).all() and (o["iters"] == 256).all()
    # beyond the table (the reference has no cap on max_iter): the kernels continue Adam's two bias corrections in double on the device --
    # 300 iterations against the C oracle, which computes them on the host for every iteration (the kernels' LONG instantiations)
    from oracle.analytic import AnalyticOracle

    a = [g[k][:16] for k in KEYS]
    want = AnalyticOracle(precision="f32").optimize(*a, 300)
    # (since round 6 the large-batch kernel has LONG instantiations too: with and without the while-condition -- never failing here -- below)
    for kernel, kw in (("w4", {}), ("auto", {}), ("w16", {}), ("w16", dict(stop_eps_pos=1e-30, stop_eps_rot=1e-30, min_loss_incr=-1e30))):
        o = opt.optimize(**{k: v[:16] for k, v in d.items()}, n_iter=300, kernel=kernel, **kw)
        err = np.linalg.norm(o["pos"].cpu().numpy() - want["pos"], axis=-1).max(axis=1) * 1000.0
        assert (o["iters"] == 300).all() and np.sort(err)[-2] <= 0.05 and err.max() <= 5.0, (kernel, err)
        np.testing.assert_allclose(o["loss"].cpu().numpy()[err <= 0.05], want["loss"][err <= 0.05], rtol=5e-3, atol=1e-6)
    with pytest.raises(_lib.DragPoserError):
        opt.optimize(**{k: v[:16] for k, v in d.items()}, n_iter=_lib.DP_MAX_ITERS + 1)
    reps = 4096  # 64 golden frames x 4096 = 262 144 frames
    big = {k: v.repeat((reps,) + (1,) * (v.dim() - 1)) for k, v in d.items()}
    for hint, kernel in ((0, "w4"), (6, "w4"), (0, "w16")):  # (the version-1 kernel hint is accepted and ignored)
        ob = opt.optimize(**big, n_iter=3, max_trackers=hint, outputs=("z", "pos", "loss"), kernel=kernel)
        os_ = opt.optimize(**d, n_iter=3, outputs=("z", "pos", "loss"), kernel=kernel)
        for k in ("z", "pos", "loss"):
            assert torch.equal(ob[k][:64], os_[k]) and torch.equal(ob[k][-64:], os_[k]), (hint, kernel, k)
            assert torch.equal(ob[k][64 * 1777:64 * 1778], os_[k])
    auto = opt.optimize(**big, n_iter=3, outputs=("z",))  # the library's choice at this size is dp_w16
    assert opt.kernel_geometry()[0] == 128 and torch.equal(auto["z"], ob["z"])


def test_untracked_frame_only_feels_the_temporal_pull(opt, dev, golden_dir):
    """Edge case: a frame with no tracker at all (E_b = 0) must not produce NaN; z moves towards z_tgt."""
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    b = {k: g[k][:16].copy() for k in KEYS}
    b["tracked"][3] = 0
    b["w"][3] = 0
    o = {k: v.cpu().numpy() for k, v in opt.optimize(**to_device_batch(b, dev), n_iter=30, lambda_tmp=0.5).items()}
    assert np.isfinite(o["z"]).all() and np.isfinite(o["pos"]).all()
    assert np.abs(o["z"][3] - b["z_tgt"][3]).max() < np.abs(b["z0"][3] - b["z_tgt"][3]).max()
    assert o["loss"][3, 0] == 0 and o["loss"][3, 1] == 0
