"""Batched latent optimiser: the host-side operator over the C ABI (include/dragposer.h).

`LatentOptimizer.optimize` runs, for B independent frames at once, what the reference's
``DragPose.run`` does per frame in its while loop (python/src/drag_pose.py:296-355): decode ->
FK -> tracker loss -> backward -> Adam on z, `n_iter` times, entirely inside one HIP kernel
launch.  PyTorch is used for device memory and streams only: tensors are handed to the library as
raw device pointers on torch's current stream.  There is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .model import DEFAULT_MODEL, NJ, HostModel

LATENT = 24

_OUT_SPECS = {  # name -> (trailing shape, dtype)
    "z": ((LATENT,), torch.float32),
    "z_pre": ((LATENT,), torch.float32),
    "pose": ((88,), torch.float32),
    "disp": ((3,), torch.float32),
    "world_disp": ((3,), torch.float32),
    "world_rot": ((4,), torch.float32),
    "pos": ((NJ, 3), torch.float32),
    "rot": ((NJ, 9), torch.float32),
    "loss": ((3,), torch.float32),
    "iters": ((), torch.int32),
    "status": ((), torch.int32),  # DP_STATUS_* bits (include/dragposer.h)
}
_LAUNCH_SPECS = {  # results that are per launch, not per frame; only on request
    "clock": ((2,), torch.int64),  # shader cycles / 100 MHz ticks of workgroup 0's iteration loop: sclk_ghz()
}


def sclk_ghz(clock):
    """the shader clock (GHz) a launch ran at, from the `clock` result (a host synchronisation)"""
    c = clock.cpu()
    return float(c[0]) / float(c[1]) * 0.1 if int(c[1]) > 0 else float("nan")


def _check(t, name, shape, dtype, device):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor on {device}")
    if t.device != device:
        raise ValueError(f"{name}: tensor is on {t.device}, the optimiser is on {device}")
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous {dtype} of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")
    return t.data_ptr()


class LaunchPlan:
    """A dp_optimize call with its arguments already checked and marshalled (LatentOptimizer.plan).  Holds the input and result
    tensors alive; `plan()` launches on torch's current stream of the optimiser's device and returns the result tensors."""

    __slots__ = ("_opt", "_b", "_p", "_r", "results", "_inputs", "_fn", "_dev")

    def __init__(self, opt, batch, params, res, tensors, inputs):
        self._opt, self.results, self._inputs = opt, tensors, inputs
        self._b, self._p, self._r = C.byref(batch), C.byref(params), C.byref(res)  # (byref objects keep their structs alive)
        self._fn, self._dev = opt.lib.dp_optimize, opt.device

    def __call__(self):
        ctx = self._opt.ctx  # (read per call: after LatentOptimizer.close() it is NULL and the library refuses, instead of a freed context being used)
        if not ctx.value:
            raise _lib.DragPoserError(_lib.DP_ERR_INVALID, "LaunchPlan: the optimiser it was made by has been closed")
        rc = self._fn(ctx, self._b, self._p, self._r, torch.cuda.current_stream(self._dev).cuda_stream)
        if rc != _lib.DP_OK:
            self._opt._fail(rc)
        return self.results


class LatentOptimizer:
    """One context per device.  Not thread-safe (same contract as the C ABI)."""

    def __init__(self, model_path=DEFAULT_MODEL, device="cuda:0", weight_dtype="fp32", arrays=None, _lib_path=None):
        self.lib = _lib.load(_lib_path)  # (_lib_path: tests only -- the second implementation kept for cross-checks)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("LatentOptimizer needs a ROCm device (cuda:N); there is no CPU fallback")
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible to PyTorch; dragposer_amd has no CPU fallback")
        self.host_model = HostModel(model_path, weight_dtype, arrays=arrays)
        self.ctx = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self.lib.dp_create(C.byref(self.ctx), C.byref(self.host_model.struct), idx)
        if rc != _lib.DP_OK:
            msg = self.lib.dp_last_error(None)
            raise _lib.DragPoserError(rc, msg.decode() if msg else "")
        fpb, tpb, lds = C.c_int(), C.c_int(), C.c_int()
        self.lib.dp_kernel_geometry(self.ctx, C.byref(fpb), C.byref(tpb), C.byref(lds))
        self.frames_per_block, self.threads_per_block, self.lds_bytes = fpb.value, tpb.value, lds.value

    def kernel_geometry(self):
        """(frames per workgroup, threads per workgroup, LDS bytes) of the kernel the last launch used"""
        fpb, tpb, lds = C.c_int(), C.c_int(), C.c_int()
        self.lib.dp_kernel_geometry(self.ctx, C.byref(fpb), C.byref(tpb), C.byref(lds))
        return fpb.value, tpb.value, lds.value

    def auto_kernel(self, n_frames):
        """"w4" | "w16": what kernel="auto" launches for a batch of `n_frames` on this device (dp_auto_kernel).  The two kernels
        differ in the last bits of their arithmetic, so a caller that cuts one batch into several launches and wants a frame's
        result not to depend on the cut pins the answer for the deciding size (dragposer_amd.sharding.pick_kernel)."""
        k = self.lib.dp_auto_kernel(self.ctx, int(n_frames))
        if k < 0:
            self._fail(k)
        return {_lib.DP_KERNEL_W4: "w4", _lib.DP_KERNEL_W16: "w16"}[k]

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.dp_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fail(self, rc):
        msg = self.lib.dp_last_error(self.ctx)
        raise _lib.DragPoserError(rc, msg.decode() if msg else "")

    def allocate_outputs(self, B, names=None):
        """a reusable set of result tensors for `optimize(..., out=...)`"""
        return {n: torch.empty((B,) + _OUT_SPECS[n][0], dtype=_OUT_SPECS[n][1], device=self.device) for n in (names or _OUT_SPECS)}

    def _outputs(self, B, names, out):
        res = _lib.DpResult()
        tensors = {}
        for name in names:
            if name in _LAUNCH_SPECS:
                shape, dtype = _LAUNCH_SPECS[name]
            else:
                shape, dtype = (B,) + _OUT_SPECS[name][0], _OUT_SPECS[name][1]
            t = out[name] if out is not None and name in out else torch.empty(shape, dtype=dtype, device=self.device)
            setattr(res, name, _check(t, name, shape, dtype, self.device))
            tensors[name] = t
        return res, tensors

    def optimize(self, z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, n_iter=50, lr=1e-2, betas=(0.9, 0.999),
                 eps=1e-8, lambda_rot=1.0, lambda_tmp=0.02, stop_eps_pos=0.0, stop_eps_rot=0.0, min_loss_incr=None,
                 max_trackers=0, outputs=None, out=None, validate_targets=False, kernel="auto", _debug=None):
        """All inputs are device tensors: z0/z_tgt [B,24], cur_rot [B,4], tgt_pos [B,22,3],
        tgt_rot [B,22,9], w [B,22,2] (fp32) and tracked [B,22] (uint8).  Returns a dict of device
        tensors (see include/dragposer.h: dp_result).  Asynchronous on torch's current stream.
        With stop_eps_* > 0 or min_loss_incr given, every frame runs the reference's own while-condition
        (drag_pose.py:300-304) and `iters` reports how many iterations it took (n_iter = max_iter).
        `max_trackers`: ignored (a kernel-selection hint of version 1; kept so that old callers keep working).
        `kernel`: "auto" | "w4" (4 frames per wave, fp32 MFMA) | "w16" (16 frames per wave, decoder on bf16 MFMA in split
        precision; what "auto" picks beyond 8192 frames: two rounds of "w4") -- include/dragposer.h: DP_KERNEL_*.
        `validate_targets`: check that every tracked joint's tgt_rot is a rotation matrix (the kernel evaluates the
        reference's |R - T|^2 in its quaternion form, equal only for orthonormal det +1 targets: include/dragposer.h) --
        costs a device reduction and a host synchronisation, so it is off by default."""
        plan = self.plan(z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, n_iter, lr, betas, eps, lambda_rot, lambda_tmp, stop_eps_pos,
                         stop_eps_rot, min_loss_incr, max_trackers, outputs, out, validate_targets, kernel)
        if _debug is not None:
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            rc = self.lib.dp_optimize_debug(self.ctx, plan._b, plan._p, plan._r, C.c_void_p(_debug.data_ptr()), stream)
            if rc != _lib.DP_OK:
                self._fail(rc)
            return plan.results
        return plan()

    def plan(self, z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked, n_iter=50, lr=1e-2, betas=(0.9, 0.999),
             eps=1e-8, lambda_rot=1.0, lambda_tmp=0.02, stop_eps_pos=0.0, stop_eps_rot=0.0, min_loss_incr=None,
             max_trackers=0, outputs=None, out=None, validate_targets=False, kernel="auto"):
        """`optimize`'s arguments checked and marshalled ONCE: returns a LaunchPlan whose call launches dp_optimize over the same
        tensors (read at launch time: refill them in place between calls) into the same result tensors, on torch's current stream --
        a caller that steps the same buffers every frame pays one ctypes call per launch instead of the checks and struct filling."""
        B = int(z0.shape[0])
        dev = self.device
        if validate_targets:
            check_rotation_targets(tgt_rot, tracked)
        batch = _lib.DpBatch()
        batch.n_frames = B
        batch.z0 = _check(z0, "z0", (B, LATENT), torch.float32, dev)
        batch.z_tgt = _check(z_tgt, "z_tgt", (B, LATENT), torch.float32, dev)
        batch.cur_rot = _check(cur_rot, "cur_rot", (B, 4), torch.float32, dev)
        batch.tgt_pos = _check(tgt_pos, "tgt_pos", (B, NJ, 3), torch.float32, dev)
        batch.tgt_rot = _check(tgt_rot, "tgt_rot", (B, NJ, 9), torch.float32, dev)
        batch.w = _check(w, "w", (B, NJ, 2), torch.float32, dev)
        batch.tracked = _check(tracked, "tracked", (B, NJ), torch.uint8, dev)
        early = min_loss_incr is not None or stop_eps_pos > 0 or stop_eps_rot > 0
        p = _lib.DpParams(n_iter=int(n_iter), lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, lambda_rot=lambda_rot,
                          lambda_tmp=lambda_tmp, early_stop=int(early), stop_eps_pos=stop_eps_pos, stop_eps_rot=stop_eps_rot,
                          min_loss_incr=float("-inf") if min_loss_incr is None else min_loss_incr,
                          max_trackers=int(max_trackers),
                          kernel={"auto": _lib.DP_KERNEL_AUTO, "w4": _lib.DP_KERNEL_W4, "w16": _lib.DP_KERNEL_W16}[kernel])
        names = tuple(outputs) if outputs is not None else tuple(_OUT_SPECS)
        res, tensors = self._outputs(B, names, out)
        return LaunchPlan(self, batch, p, res, tensors, (z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked))

    def forward(self, z, cur_rot, outputs=("pose", "disp", "world_disp", "world_rot", "pos", "rot"), out=None):
        """decode + FK of z [B,24] under cur_rot [B,4] (no loss, no update)."""
        B = int(z.shape[0])
        zp = _check(z, "z", (B, LATENT), torch.float32, self.device)
        cp = _check(cur_rot, "cur_rot", (B, 4), torch.float32, self.device)
        res, tensors = self._outputs(B, tuple(outputs), out)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self.lib.dp_forward(self.ctx, B, C.c_void_p(zp), C.c_void_p(cp), C.byref(res), stream)
        if rc != _lib.DP_OK:
            self._fail(rc)
        return tensors


    def sequence_advance(self, frame, global_pos, global_rot, latent_buf, disp_buf, heights_buf, height_joints, pose_ret=None,
                         pos_ret=None, adjust=None, tgt_pos=None):
        """The reference's per-frame epilogue (drag_pose.py:369-402) for S sequences in one launch: updates
        `global_pos`, `global_rot` and the three history buffers IN PLACE from `frame` (the dict `optimize` returned:
        z_pre, pose, disp, world_disp, world_rot, pos) and fills `pose_ret` / `pos_ret`.
        `adjust` = (joint, target_joint, weight) or None; `tgt_pos` = this frame's dense [S,22,3] targets."""
        S = int(global_pos.shape[0])
        dev = self.device
        res = _lib.DpResult()
        for name in ("z_pre", "pose", "disp", "world_disp", "world_rot", "pos"):
            shape, dtype = _OUT_SPECS[name]
            setattr(res, name, _check(frame[name], name, (S,) + shape, dtype, dev))
        H, NH = int(latent_buf.shape[1]), len(height_joints)
        st = _lib.DpSeqState()
        st.global_pos = _check(global_pos, "global_pos", (S, 3), torch.float32, dev)
        st.global_rot = _check(global_rot, "global_rot", (S, 4), torch.float32, dev)
        st.latent_buf = _check(latent_buf, "latent_buf", (S, H, LATENT), torch.float32, dev)
        st.disp_buf = _check(disp_buf, "disp_buf", (S, H, 3), torch.float32, dev)
        st.heights_buf = _check(heights_buf, "heights_buf", (S, H, NH), torch.float32, dev)
        st.history, st.n_heights = H, NH
        for i, j in enumerate(height_joints):
            st.height_joints[i] = int(j)
        step = _lib.DpSeqStep()
        step.adjust_joint = -1
        if adjust is not None:
            step.adjust_joint, step.adjust_target_joint, step.adjust_weight = int(adjust[0]), int(adjust[1]), float(adjust[2])
            step.tgt_pos = _check(tgt_pos, "tgt_pos", (S, NJ, 3), torch.float32, dev)
        if pose_ret is not None:
            step.pose_ret = _check(pose_ret, "pose_ret", (S, 88), torch.float32, dev)
        if pos_ret is not None:
            step.pos_ret = _check(pos_ret, "pos_ret", (S, 3), torch.float32, dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self.lib.dp_sequence_advance(self.ctx, S, C.byref(res), C.byref(st), C.byref(step), stream)
        if rc != _lib.DP_OK:
            self._fail(rc)


def _optimize_sequence(self, latent, tgt_pos, tgt_rot, tgt_root, w, tracked, z_tgt, z_tgt_strides, global_pos, global_rot, latent_buf, disp_buf,
                       heights_buf, height_joints, n_iter=100, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, lambda_rot=1.0, lambda_tmp=0.0,
                       stop_eps_pos=1e-4, stop_eps_rot=1e-2, min_loss_incr=1e-5, adjust=None, pose_ret=None, pos_ret=None, iters=None,
                       loss=None, scratch=None, status=None):
    """T consecutive frames of S sequences in one launch (include/dragposer.h: dp_optimize_sequence): the optimise loop with the
    reference's while-condition and run()'s epilogue per frame, state carried on the device.  tgt_pos [T,S,22,3] / tgt_rot
    [T,S,22,9] dense per joint; tgt_root [T,S,3] or None (position targets are then tgt_pos + (tgt_root[t] - running global
    position), eval_drag.py:186-199); w [S,22,2], tracked [S,22]; z_tgt any fp32 device tensor addressed with `z_tgt_strides` =
    (floats between steps, floats between sequences).  `latent` [S,24], `global_pos`, `global_rot` and the three history
    buffers are updated IN PLACE.  Returns dict(pose_ret [T,S,88], pos_ret [T,S,3], iters [T,S], loss [T,S,3], status [T,S]: DP_STATUS_* bits)."""
    T, S = int(tgt_pos.shape[0]), int(tgt_pos.shape[1])
    dev = self.device
    H, NH = int(latent_buf.shape[1]), len(height_joints)
    fr = _lib.DpSeqFrames()
    fr.n_steps = T
    fr.tgt_pos = _check(tgt_pos, "tgt_pos", (T, S, NJ, 3), torch.float32, dev)
    fr.tgt_rot = _check(tgt_rot, "tgt_rot", (T, S, NJ, 9), torch.float32, dev)
    fr.tgt_root = _check(tgt_root, "tgt_root", (T, S, 3), torch.float32, dev) if tgt_root is not None else None
    fr.w = _check(w, "w", (S, NJ, 2), torch.float32, dev)
    fr.tracked = _check(tracked, "tracked", (S, NJ), torch.uint8, dev)
    if z_tgt.device != dev or z_tgt.dtype != torch.float32:
        raise ValueError("z_tgt: expected an fp32 tensor on the optimiser's device")
    fr.z_tgt, fr.z_tgt_step, fr.z_tgt_seq = z_tgt.data_ptr(), int(z_tgt_strides[0]), int(z_tgt_strides[1])
    st = _lib.DpSeqState()
    st.global_pos = _check(global_pos, "global_pos", (S, 3), torch.float32, dev)
    st.global_rot = _check(global_rot, "global_rot", (S, 4), torch.float32, dev)
    st.latent_buf = _check(latent_buf, "latent_buf", (S, H, LATENT), torch.float32, dev)
    st.disp_buf = _check(disp_buf, "disp_buf", (S, H, 3), torch.float32, dev)
    st.heights_buf = _check(heights_buf, "heights_buf", (S, H, NH), torch.float32, dev)
    st.history, st.n_heights = H, NH
    for i, j in enumerate(height_joints):
        st.height_joints[i] = int(j)
    step = _lib.DpSeqStep()
    step.adjust_joint = -1
    if adjust is not None:
        step.adjust_joint, step.adjust_target_joint, step.adjust_weight = int(adjust[0]), int(adjust[1]), float(adjust[2])
    res = _lib.DpSeqResults()
    outs = {}
    res.world_rot = None
    for name, t, shape, dtype in (("pose_ret", pose_ret, (T, S, 88), torch.float32), ("pos_ret", pos_ret, (T, S, 3), torch.float32),
                                  ("iters", iters, (T, S), torch.int32), ("loss", loss, (T, S, 3), torch.float32),
                                  ("status", status, (T, S), torch.int32)):
        t = t if t is not None else torch.empty(shape, dtype=dtype, device=dev)
        setattr(res, name, _check(t, name, shape, dtype, dev))
        outs[name] = t
    scratch = scratch if scratch is not None else torch.empty(T, S, LATENT + 3 + NH, device=dev)
    res.hist_scratch = _check(scratch, "scratch", (T, S, LATENT + 3 + NH), torch.float32, dev)
    p = _lib.DpParams(n_iter=int(n_iter), lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, lambda_rot=lambda_rot, lambda_tmp=lambda_tmp,
                      early_stop=1, stop_eps_pos=stop_eps_pos, stop_eps_rot=stop_eps_rot,
                      min_loss_incr=float("-inf") if min_loss_incr is None else min_loss_incr, max_trackers=0, kernel=_lib.DP_KERNEL_AUTO)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = self.lib.dp_optimize_sequence(self.ctx, S, C.c_void_p(_check(latent, "latent", (S, LATENT), torch.float32, dev)), C.byref(fr),
                                       C.byref(p), C.byref(st), C.byref(step), C.byref(res), stream)
    if rc != _lib.DP_OK:
        self._fail(rc)
    return outs


LatentOptimizer.optimize_sequence = _optimize_sequence


def check_rotation_targets(tgt_rot, tracked, tol=1e-3):
    """Raises ValueError unless every tracked joint's 3x3 target is orthonormal with determinant +1 (within `tol`)."""
    R = tgt_rot.reshape(-1, NJ, 3, 3)
    m = tracked.reshape(-1, NJ).bool()
    if not bool(m.any()):
        return
    Rt = R[m].double()
    dev_orth = (Rt @ Rt.transpose(-1, -2) - torch.eye(3, dtype=Rt.dtype, device=Rt.device)).abs().amax()
    dev_det = (torch.linalg.det(Rt) - 1.0).abs().amax()
    worst = float(torch.maximum(dev_orth, dev_det))
    if not worst <= tol:
        raise ValueError(f"tgt_rot: tracked targets must be rotation matrices (orthonormal, det +1); worst deviation {worst:.3g} > {tol}")


def to_device_batch(np_batch, device):
    """numpy dict (z0, z_tgt, cur_rot, tgt_pos, tgt_rot, w, tracked) -> device tensors."""
    out = {}
    for k in ("z0", "z_tgt", "cur_rot", "tgt_pos", "tgt_rot", "w"):
        out[k] = torch.from_numpy(np.ascontiguousarray(np_batch[k], dtype=np.float32)).to(device)
    out["tracked"] = torch.from_numpy(np.ascontiguousarray(np_batch["tracked"], dtype=np.uint8)).to(device)
    return out
