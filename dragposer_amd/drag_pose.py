"""`DragPose`: the reference's stateful per-frame operator (python/src/drag_pose.py:12-414) for S
sequences advancing in lock-step, with the optimise loop dispatched to the HIP kernel.

What stays here (PyTorch-ROCm device tensors, a handful of tiny ops per frame):
  * the state the reference keeps between frames: latent, current global position /
    rotation, the 60-deep latent / displacement / height ring buffers, the temporal target buffer;
  * the temporal target block (row a12, drag_pose.py:234-294), which calls the Transformer;
  * the per-frame epilogue (row a11, drag_pose.py:369-402): global pose update, joint adjustment,
    buffer shifts, root channels of the returned pose.
What does not: decode / FK / loss / backward / Adam / the while-condition -- those run inside
`LatentOptimizer.optimize` (one kernel launch per frame index for all S sequences).

`run()` keeps the reference's argument names and meaning; tensors carry a leading sequence
dimension S (a single sequence may omit it, then the results omit it too, like the reference).
"""
import torch

from .model import NJ
from .optimizer import LATENT, LatentOptimizer
from .temporal import HISTORY, PAST_FRAMES, SAMPLE_STEP


class DragPose:
    def __init__(self, optimizer: LatentOptimizer, temporal_model, means_latent, stds_latent, n_sequences=1):
        self.opt = optimizer
        self.device = optimizer.device
        self.temporal = temporal_model.to(self.device).eval() if temporal_model is not None else None
        self.S = int(n_sequences)
        dev = self.device
        hm = optimizer.host_model.arrays
        self.means_q = torch.from_numpy(hm["mean_q"]).to(dev)  # drag_pose.py:27-33
        self.stds_q = torch.from_numpy(hm["std_q"]).to(dev)
        self.offsets = torch.from_numpy(hm["offsets"]).to(dev)
        self.means_latent = torch.as_tensor(means_latent, dtype=torch.float32, device=dev)
        self.stds_latent = torch.as_tensor(stds_latent, dtype=torch.float32, device=dev)
        self.temporal_frames_index = list(PAST_FRAMES)
        self.target_latent_buffer = None
        self.latent = None
        self.last = None
        self._idx_cache = {}

    def _index(self, values):
        """device index tensor for a Python list, created once (no host->device copy inside a captured step)"""
        key = tuple(int(v) for v in values)
        if key not in self._idx_cache:
            self._idx_cache[key] = torch.tensor(key, dtype=torch.int64, device=self.device)
        return self._idx_cache[key]

    # ------------------------------------------------------------------ state (drag_pose.py:47-64)
    def set_initial_state(self, latent, init_global_pos, initial_global_rot, initial_heights):
        """What set_initial_pose leaves behind, with the initial latent given instead of encoded."""
        dev, S = self.device, self.S
        f = lambda t, shape: torch.as_tensor(t, dtype=torch.float32, device=dev).reshape(shape).clone()
        self.latent = f(latent, (S, LATENT))
        self.current_global_pos = f(init_global_pos, (S, 3))
        self.current_global_rot = f(initial_global_rot, (S, 4))
        self.latent_buffer = self.latent.unsqueeze(1).repeat(1, HISTORY, 1)
        self.displacement_buffer = torch.zeros(S, HISTORY, 3, device=dev)
        self.heights_buffer = f(initial_heights, (S, 1, -1)).repeat(1, HISTORY, 1)
        self.current_index = 0
        self.target_latent_buffer = None

    # ------------------------------------------------------------------ temporal target (drag_pose.py:234-294)
    def _temporal_targets(self, window):
        S, dev = self.S, self.device
        assert window % SAMPLE_STEP == 0
        if self.target_latent_buffer is None or self.target_latent_buffer.shape[1] != window + 1:
            self.target_latent_buffer = torch.zeros(S, window + 1, LATENT, device=dev)
        if self.current_index != 0 or self.temporal is None:  # no predictor: the buffer stays zero, use lambda_temporal = 0
            return
        idx = self.temporal_frames_index
        idx_t = self._index(idx)
        with torch.no_grad():
            input_latent = self.latent_buffer.index_select(1, idx_t)[:, :-1].clone()
            input_disp = torch.stack([self.displacement_buffer[:, j:j + SAMPLE_STEP].sum(dim=1) for j in idx[:-1]], dim=1)
            tgt = self.latent_buffer[:, idx[-1]].unsqueeze(1).clone()
            input_latent = (input_latent - self.means_latent) / self.stds_latent
            tgt = (tgt - self.means_latent) / self.stds_latent
            heights = self.heights_buffer.index_select(1, idx_t)[:, :-1].clone()
            enc_in = torch.cat((input_latent, input_disp, heights), dim=-1)
            buf = self.target_latent_buffer
            for i in range(0, window + 1, SAMPLE_STEP):
                pred = self.temporal(enc_in, tgt)
                tgt = torch.cat((tgt, pred[:, -1:]), dim=1)
                buf[:, i] = pred[:, -1]
            buf = buf * self.stds_latent + self.means_latent
            for i in range(0, window, SAMPLE_STEP):  # "lerp" with weight 1: hold the next prediction
                buf[:, i:i + SAMPLE_STEP + 1] = buf[:, i + SAMPLE_STEP].unsqueeze(1)
            self.target_latent_buffer = buf

    # ------------------------------------------------------------------ one frame (drag_pose.py:196-414)
    def run(self, target_ee_pos, target_ee_rot, mask_joints, weights_joints, offsets=None, stop_eps_pos=1e-2,
            stop_eps_rot=1e-2, max_iter=100, min_loss_incr=0.00001, learning_rate=1e-3, lambda_rot=1, lambda_temporal=1,
            temporal_future_window=60, height_indices=(0, 4, 8, 13, 17, 21), joint_adjustment_indices=None,
            joint_adjustment_weight=0.01, verbose=False):
        dev, S = self.device, self.S
        squeeze = torch.as_tensor(target_ee_pos).dim() == 2
        tp = torch.as_tensor(target_ee_pos, dtype=torch.float32, device=dev).reshape(S, -1, 3)
        tR = torch.as_tensor(target_ee_rot, dtype=torch.float32, device=dev).reshape(S, -1, 9)
        mj = torch.as_tensor(mask_joints, dtype=torch.int64, device=dev).reshape(-1)
        wj = torch.as_tensor(weights_joints, dtype=torch.float32, device=dev).reshape(-1, 2)
        E = mj.numel()
        if tp.shape[1] != E or tR.shape[1] != E or wj.shape[0] != E:
            raise ValueError("target_ee_pos / target_ee_rot / weights_joints must have one row per entry of mask_joints")
        if offsets is not None and not torch.allclose(torch.as_tensor(offsets, dtype=torch.float32, device=dev).reshape(NJ, 3),
                                                      self.offsets, atol=1e-6):
            raise ValueError("offsets differ from the skeleton the optimiser context was created with")

        self._temporal_targets(temporal_future_window)
        target_latent = self.target_latent_buffer[:, self.current_index].contiguous()

        # scatter the E tracker rows to dense per-joint arrays (the C ABI's layout)
        tgt_pos = torch.zeros(S, NJ, 3, device=dev)
        tgt_rot = torch.zeros(S, NJ, 9, device=dev)
        w = torch.zeros(S, NJ, 2, device=dev)
        tracked = torch.zeros(S, NJ, dtype=torch.uint8, device=dev)
        tgt_pos.index_copy_(1, mj, tp)
        tgt_rot.index_copy_(1, mj, tR)
        w.index_copy_(1, mj, wj.unsqueeze(0).expand(S, -1, -1).contiguous())
        tracked.index_fill_(1, mj, 1)

        out = self.opt.optimize(self.latent.contiguous(), target_latent, self.current_global_rot.contiguous(), tgt_pos, tgt_rot,
                                w, tracked, n_iter=max_iter, lr=learning_rate, lambda_rot=float(lambda_rot),
                                lambda_tmp=float(lambda_temporal), stop_eps_pos=stop_eps_pos, stop_eps_rot=stop_eps_rot,
                                min_loss_incr=min_loss_incr, max_trackers=E)
        self.last = out
        self.latent = out["z"]
        if verbose:
            l, it = out["loss"].cpu(), out["iters"].cpu()
            print(f"Loss sqrt(Pos): {l[:, 0].sqrt().mean():.5f} // Loss Rot: {l[:, 1].mean():.5f} // "
                  f"Loss Temporal: {l[:, 2].mean():.5f} // Iter: {it.float().mean():.1f}")

        # ---- epilogue (drag_pose.py:369-402)
        self.current_global_pos = self.current_global_pos + out["world_disp"]
        self.current_global_rot = out["world_rot"]
        displacement = out["disp"].clone()
        if joint_adjustment_indices is not None:
            joint_index, ee_index = joint_adjustment_indices
            adj = (tp[:, ee_index] - out["pos"][:, joint_index]) * joint_adjustment_weight
            self.current_global_pos = self.current_global_pos + adj
            displacement = displacement + adj
        self.latent_buffer = torch.cat((self.latent_buffer[:, 1:], out["z_pre"].unsqueeze(1)), dim=1)
        self.displacement_buffer = torch.cat((self.displacement_buffer[:, 1:], displacement.unsqueeze(1)), dim=1)
        heights = (out["pos"] + self.current_global_pos.unsqueeze(1)).index_select(1, self._index(height_indices))[:, :, 1]
        self.heights_buffer = torch.cat((self.heights_buffer[:, 1:], heights.unsqueeze(1)), dim=1)
        pose = out["pose"].clone()
        pose[:, :4] = (self.current_global_rot - self.means_q[:4]) / self.stds_q[:4]
        self.current_index = 0 if temporal_future_window == 0 else (self.current_index + 1) % temporal_future_window
        if squeeze and S == 1:
            return pose[0], self.current_global_pos[0]
        return pose, self.current_global_pos
