"""GPU: the C ABI from a plain C program (tests/c_caller.c, compiled here with gcc against include/dragposer.h and linked to
dragposer_amd/lib/libdragposer_hip.so -- no HIP headers, no torch, device buffers through the library's dp_io_* helpers): the same frames through
`dp_create -> dp_io_upload -> dp_optimize -> dp_io_download` as the Python operator runs them, bit for bit; a struct of the wrong size refused."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("early", [0, 1])
def test_plain_c_caller_equals_the_python_operator(tmp_path, early):
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    from dragposer_amd.model import HostModel
    from dragposer_amd.optimizer import LatentOptimizer, to_device_batch

    libdir = os.path.join(ROOT, "dragposer_amd", "lib")
    exe = tmp_path / "c_caller"
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), os.path.join(ROOT, "tests", "c_caller.c"),
                           "-L", libdir, "-ldragposer_hip", "-Wl,-rpath," + libdir])
    hm = HostModel()
    for k, v in hm.arrays.items():
        np.ascontiguousarray(v, np.float32).tofile(tmp_path / f"{k}.bin")
    hm.parents.astype(np.int32).tofile(tmp_path / "parents.bin")
    B, n_iter = 200, 30  # (a ragged tail: 200 = 12 workgroups of 16 frames + 8)
    b = R.synth_inputs(R.OracleModel(), B, seed=31)
    for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w"):
        np.ascontiguousarray(b[k], np.float32).tofile(tmp_path / f"{k}.bin")
    np.ascontiguousarray(b["tracked"], np.uint8).tofile(tmp_path / "tracked.bin")
    # (one HIP runtime per process and the library binds to the first one it finds: the child gets torch's, as the Python binding does)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([str(exe), str(tmp_path), str(B), str(n_iter), str(early)], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "16 frames / 256 threads" in p.stdout, p.stdout
    opt = LatentOptimizer(device="cuda:0")
    kw = dict(stop_eps_pos=1e-4, stop_eps_rot=1e-2, min_loss_incr=1e-5) if early else {}
    want = opt.optimize(**to_device_batch(b, opt.device), n_iter=n_iter, lambda_tmp=0.02, **kw)
    torch.cuda.synchronize()
    for name, key, dt in (("out_z", "z", np.float32), ("out_pos", "pos", np.float32), ("out_pose", "pose", np.float32), ("out_loss", "loss", np.float32),
                          ("out_iters", "iters", np.int32), ("out_status", "status", np.int32)):
        got = np.fromfile(tmp_path / f"{name}.bin", dtype=dt).reshape(want[key].shape)
        np.testing.assert_array_equal(got, want[key].cpu().numpy(), err_msg=name)
    assert (np.fromfile(tmp_path / "out_status.bin", dtype=np.int32) == 0).all()
