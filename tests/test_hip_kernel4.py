"""GPU (MI355X): the two-workgroups-per-CU kernel (dp_kernel4.hip) against the 8-wave kernel (dp_kernel.hip).

Both run the same arithmetic in the same order, so the bar is BIT equality on every output; parity of the
8-wave kernel with the oracle / goldens is tests/test_hip_parity.py.  The library picks the kernel from the
caller's `max_trackers` hint; DP_KERNEL in the environment forces one (read at every launch).
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

KEYS = ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w", "tracked")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def opt(dev):
    from dragposer_amd.optimizer import LatentOptimizer

    return LatentOptimizer(device=dev)


class forced:
    """DP_KERNEL=<which> for the launches inside the block"""

    def __init__(self, which):
        self.which = which

    def __enter__(self):
        self.prev = os.environ.get("DP_KERNEL")
        os.environ["DP_KERNEL"] = self.which

    def __exit__(self, *exc):
        if self.prev is None:
            os.environ.pop("DP_KERNEL", None)
        else:
            os.environ["DP_KERNEL"] = self.prev


def _run(o, d, which, expect_threads, **kw):
    with forced(which):
        out = o.optimize(**d, **kw)
        torch.cuda.synchronize()
    assert o.kernel_geometry()[1] == expect_threads, (which, o.kernel_geometry())
    return {k: v.cpu().numpy() for k, v in out.items()}


def _same(a, b):
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("which", ["4x1", "4x2"])
@pytest.mark.parametrize("B", [1, 7, 8, 9, 16, 17, 33, 64])
def test_bit_equal_to_the_8_wave_kernel(opt, dev, golden_dir, which, B):
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    d = to_device_batch({k: g[k][:B] for k in KEYS}, dev)
    ref = _run(opt, d, "8", 512, n_iter=20, max_trackers=6)
    got = _run(opt, d, which, 256, n_iter=20, max_trackers=6)
    assert opt.kernel_geometry()[0] == (8 if which == "4x1" else 16)
    _same(got, ref)


@pytest.mark.parametrize("which", ["4x1", "4x2"])
def test_debug_dump_and_forward_bit_equal(opt, dev, golden_dir, which):
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, "s4.npz"))
    B = g["z0"].shape[0]
    d = to_device_batch(g, dev)
    dumps = {}
    for w, thr in (("8", 512), (which, 256)):
        dbg = torch.zeros(B, 240, device=dev)
        _run(opt, d, w, thr, n_iter=1, max_trackers=6, _debug=dbg)
        dumps[w] = dbg.cpu().numpy()
    np.testing.assert_array_equal(dumps[which], dumps["8"])
    assert np.abs(dumps["8"][:, 104:208]).max() > 0  # the dump is live
    with forced("8"):
        f8 = {k: v.cpu().numpy() for k, v in opt.forward(d["z0"], d["cur_rot"]).items()}
    with forced(which):
        f4 = {k: v.cpu().numpy() for k, v in opt.forward(d["z0"], d["cur_rot"]).items()}
        assert opt.kernel_geometry()[1] == 256
    _same(f4, f8)


@pytest.mark.parametrize("which", ["4x1", "4x2"])
@pytest.mark.parametrize("n_trk", [1, 3, 9, 16])
def test_tracker_counts_up_to_capacity(opt, dev, which, n_trk):
    from dragposer_amd.optimizer import to_device_batch

    m = R.OracleModel()
    b = R.synth_inputs(m, 40)
    rs = np.random.RandomState(n_trk)
    with torch.no_grad():
        mo, dd = R.decoder_forward(m, torch.tensor(b["z_src"]))
        _, _, pos, rot, _ = R.pose_fk(m, mo, dd, torch.tensor(b["cur_rot"]))
    b["tracked"][:] = 0
    b["w"][:] = 0
    for f in range(40):
        n = rs.randint(0, n_trk + 1) if f % 4 == 3 else n_trk  # ragged tracker counts, incl. untracked frames
        js = np.sort(rs.permutation(22)[:n])
        b["tracked"][f, js] = 1
        b["w"][f, js, 0] = rs.uniform(1, 10, n)
        b["w"][f, js, 1] = rs.uniform(0.01, 2, n)
    trk = b["tracked"].astype(bool)[..., None]
    b["tgt_pos"] = (pos.numpy() * trk).astype(np.float32)
    b["tgt_rot"] = (rot.numpy().reshape(40, 22, 9) * trk).astype(np.float32)
    d = to_device_batch(b, dev)
    ref = _run(opt, d, "8", 512, n_iter=25, max_trackers=n_trk)
    got = _run(opt, d, which, 256, n_iter=25, max_trackers=n_trk)
    _same(got, ref)


def test_dispatch_rules(opt, dev, golden_dir):
    """no hint, a hint above the capacity, or early stop -> the 8-wave kernel, whatever DP_KERNEL asks for"""
    from dragposer_amd.optimizer import to_device_batch

    g = R.load_golden(os.path.join(golden_dir, "s1.npz"))
    d = to_device_batch(g, dev)
    os.environ.pop("DP_KERNEL", None)
    opt.optimize(**d, n_iter=2)
    assert opt.kernel_geometry()[:2] == (16, 512)
    opt.optimize(**d, n_iter=2, max_trackers=6)
    assert opt.kernel_geometry()[:2] == (16, 512)  # small batch: one workgroup per CU at most, nothing to co-schedule
    opt.optimize(**d, n_iter=2, max_trackers=22)
    assert opt.kernel_geometry()[:2] == (16, 512)
    opt.optimize(**d, n_iter=5, max_trackers=6, stop_eps_pos=1e-2, stop_eps_rot=1e-2, min_loss_incr=1e-5)
    assert opt.kernel_geometry()[:2] == (16, 512)
    with forced("4x2"):
        opt.optimize(**d, n_iter=2)  # no hint: the override is not honoured
        assert opt.kernel_geometry()[:2] == (16, 512)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B", [4096, 8192 + 24])
def test_full_size_batches_bit_equal(opt, dev, B):
    from dragposer_amd.optimizer import to_device_batch

    b = R.synth_inputs(R.OracleModel(), B)
    d = to_device_batch(b, dev)
    ref = _run(opt, d, "8", 512, n_iter=50, max_trackers=6)
    for which in ("4x1", "4x2"):
        _same(_run(opt, d, which, 256, n_iter=50, max_trackers=6), ref)
    os.environ.pop("DP_KERNEL", None)
    auto = opt.optimize(**d, n_iter=50, max_trackers=6)
    assert opt.kernel_geometry()[:2] == ((16, 512) if B <= 4096 else (16, 256))
    np.testing.assert_array_equal(auto["z"].cpu().numpy(), ref["z"])


def test_understated_hint_is_memory_safe(opt, dev):
    """a caller that states max_trackers = 6 but tracks all 22 joints gets the first 16 (the kernel's capacity) used and the
    rest ignored -- documented, finite, and no out-of-bounds access"""
    from dragposer_amd.optimizer import to_device_batch

    b = R.synth_inputs(R.OracleModel(), 9000)
    b["tracked"][:] = 1
    b["w"][:] = 1.0
    d = to_device_batch(b, dev)
    os.environ.pop("DP_KERNEL", None)
    o = opt.optimize(**d, n_iter=5, max_trackers=6)
    torch.cuda.synchronize()
    assert opt.kernel_geometry()[:2] == (16, 256)
    assert all(torch.isfinite(v).all() for k, v in o.items() if v.dtype == torch.float32)
