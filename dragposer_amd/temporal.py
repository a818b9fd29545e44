"""The temporal predictor, kept in PyTorch (ROCm) as the reference keeps it.

`TemporalPredictor` has the reference's architecture and state_dict layout
(python/src/temporal_transformer.py:7-77, positional_encoding.py:6-32; hyper-parameters
train_temporal.py:17-37), so a ``temporal.pt`` trained with the reference loads unchanged via
`load_reference_checkpoint`.  It produces `target_latent`, the anchor of the lambda_temporal pull
term.  `NativeTemporal` runs the same network (and the whole temporal target block of DragPose.run,
drag_pose.py:248-292) in one HIP launch through the C ABI (dp_temporal_*, csrc/dp_temporal.hip): what the native
Unity plug-in uses, where there is no PyTorch.
"""
import math

import torch
import torch.nn as nn

SAMPLE_STEP = 4                                   # train_temporal.py:15
PAST_FRAMES = list(range(0, 60, SAMPLE_STEP))     # train_temporal.py:22
FUTURE_FRAMES = list(range(60, 120, SAMPLE_STEP)) # train_temporal.py:23
HISTORY = FUTURE_FRAMES[0]                        # ring-buffer depth (drag_pose.py:37-41)


class _PositionalEncoding(nn.Module):
    def __init__(self, dim_model, dropout_p, max_len):
        super().__init__()
        self.dropout = nn.Dropout(dropout_p)
        pe = torch.zeros(max_len, dim_model)
        pos = torch.arange(0, max_len, dtype=torch.float).view(-1, 1)
        div = torch.exp(torch.arange(0, dim_model, 2).float() * (-math.log(10000.0)) / dim_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pos_encoding", pe)

    def forward(self, x):
        return self.dropout(x + self.pos_encoding[: x.size(1), :])


class TemporalPredictor(nn.Module):
    def __init__(self, latent_dim=24, n_heights=6, n_heads=4, n_encoder_layers=3, n_decoder_layers=3,
                 dim_feedforward=2048, dropout=0.1):
        super().__init__()
        d_model = latent_dim * 2
        self.in_dropout = nn.Dropout(dropout)
        self.positional_encoding = _PositionalEncoding(d_model, dropout, len(PAST_FRAMES) + len(FUTURE_FRAMES))
        self.in_proj_encoder = nn.Linear(latent_dim + 3 + n_heights, d_model)
        self.in_proj_decoder = nn.Linear(latent_dim, d_model)
        self.temporal = nn.Transformer(d_model=d_model, nhead=n_heads, num_encoder_layers=n_encoder_layers,
                                       num_decoder_layers=n_decoder_layers, dim_feedforward=dim_feedforward, dropout=dropout)
        self.out_proj = nn.Linear(d_model, latent_dim)

    def forward(self, latent, latent_target, tgt_mask=None):
        """latent [S, past, 33], latent_target [S, future, 24] -> [S, future, 24]"""
        x = self.positional_encoding(self.in_proj_encoder(self.in_dropout(latent))).permute(1, 0, 2)
        y = self.positional_encoding(self.in_proj_decoder(latent_target)).permute(1, 0, 2)
        return self.out_proj(self.temporal(x, y, tgt_mask=tgt_mask)).permute(1, 0, 2)


def load_reference_checkpoint(path, device="cpu", **arch):
    """temporal.pt as saved by the reference (train.py:308-318): model_state_dict + latent statistics."""
    ck = torch.load(path, map_location=device)
    sd = ck["model_state_dict"]
    # layer counts and feed-forward width from the tensors themselves (train_temporal.param holds them in the reference: 3 / 3 / 2048)
    n_layers = lambda part: 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(f"temporal.{part}.layers."))
    arch = dict(dict(n_encoder_layers=n_layers("encoder"), n_decoder_layers=n_layers("decoder"),
                     dim_feedforward=int(sd["temporal.encoder.layers.0.linear1.weight"].shape[0])), **arch)
    model = TemporalPredictor(**arch).to(device)
    model.load_state_dict(ck["model_state_dict"])
    return model.eval(), ck["means_latent"].to(device), ck["stds_latent"].to(device)


class NativeTemporal:
    """The device-side predictor behind `dp_temporal_create / dp_temporal_predict` (include/dragposer.h), built from a
    `TemporalPredictor` (or any module / dict with the reference's state_dict layout) and the latent statistics."""

    def __init__(self, model, means_latent, stds_latent, device="cuda:0", sample_step=SAMPLE_STEP):
        import ctypes as C
        import numpy as np
        from . import _lib

        sd = model.state_dict() if hasattr(model, "state_dict") else dict(model)
        keep = []  # host arrays the structs point into, alive until dp_temporal_create has copied them

        def arr(v):
            a = np.ascontiguousarray((v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)), dtype=np.float32)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_float))

        def layers(prefix, n, dec):
            L = (_lib.DpTemporalLayer * n)()
            for i in range(n):
                p = f"{prefix}.layers.{i}."
                L[i].sa_in_w, L[i].sa_in_b = arr(sd[p + "self_attn.in_proj_weight"]), arr(sd[p + "self_attn.in_proj_bias"])
                L[i].sa_out_w, L[i].sa_out_b = arr(sd[p + "self_attn.out_proj.weight"]), arr(sd[p + "self_attn.out_proj.bias"])
                if dec:
                    L[i].ca_in_w, L[i].ca_in_b = arr(sd[p + "multihead_attn.in_proj_weight"]), arr(sd[p + "multihead_attn.in_proj_bias"])
                    L[i].ca_out_w, L[i].ca_out_b = arr(sd[p + "multihead_attn.out_proj.weight"]), arr(sd[p + "multihead_attn.out_proj.bias"])
                    L[i].norm3_w, L[i].norm3_b = arr(sd[p + "norm3.weight"]), arr(sd[p + "norm3.bias"])
                L[i].lin1_w, L[i].lin1_b = arr(sd[p + "linear1.weight"]), arr(sd[p + "linear1.bias"])
                L[i].lin2_w, L[i].lin2_b = arr(sd[p + "linear2.weight"]), arr(sd[p + "linear2.bias"])
                L[i].norm1_w, L[i].norm1_b = arr(sd[p + "norm1.weight"]), arr(sd[p + "norm1.bias"])
                L[i].norm2_w, L[i].norm2_b = arr(sd[p + "norm2.weight"]), arr(sd[p + "norm2.bias"])
            return L

        n_enc = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("temporal.encoder.layers."))
        n_dec = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("temporal.decoder.layers."))
        m = _lib.DpTemporalModel()
        m.n_heights = int(sd["in_proj_encoder.weight"].shape[1]) - 24 - 3
        m.dim_feedforward = int(sd["temporal.encoder.layers.0.linear1.weight"].shape[0])
        m.n_encoder_layers, m.n_decoder_layers = n_enc, n_dec
        m.max_len = int(sd["positional_encoding.pos_encoding"].shape[0])
        m.sample_step = int(sample_step)
        m.in_proj_encoder_w, m.in_proj_encoder_b = arr(sd["in_proj_encoder.weight"]), arr(sd["in_proj_encoder.bias"])
        m.in_proj_decoder_w, m.in_proj_decoder_b = arr(sd["in_proj_decoder.weight"]), arr(sd["in_proj_decoder.bias"])
        m.out_proj_w, m.out_proj_b = arr(sd["out_proj.weight"]), arr(sd["out_proj.bias"])
        m.pos_encoding = arr(sd["positional_encoding.pos_encoding"])
        m.enc_norm_w, m.enc_norm_b = arr(sd["temporal.encoder.norm.weight"]), arr(sd["temporal.encoder.norm.bias"])
        m.dec_norm_w, m.dec_norm_b = arr(sd["temporal.decoder.norm.weight"]), arr(sd["temporal.decoder.norm.bias"])
        m.means_latent, m.stds_latent = arr(means_latent), arr(stds_latent)
        enc, dec = layers("temporal.encoder", n_enc, False), layers("temporal.decoder", n_dec, True)
        m.enc, m.dec = enc, dec
        self._lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("NativeTemporal runs on an MI355X only (there is no CPU fallback)")
        if self.device.index is None:  # "cuda" = the current device, as LatentOptimizer resolves it
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_heights = m.n_heights
        h = C.c_void_p()
        rc = self._lib.dp_temporal_create(C.byref(h), C.byref(m), self.device.index)
        if rc != _lib.DP_OK:
            msg = self._lib.dp_temporal_last_error(None)
            raise _lib.DragPoserError(rc, msg.decode() if msg else "")
        self._h = h

    def _force_variant(self, variant):
        """tests only: pin the kernel variant (21 / 41 / 42: waves per SIMD, sequences per workgroup; 102 / 104 / 108 / 116: a team of that many
        workgroups per sequence; 0: chosen from the batch)"""
        rc = self._lib.dp_temporal_debug_force_variant(self._h, int(variant))
        if rc != 0:
            raise ValueError(f"dp_temporal_debug_force_variant({variant}) -> {rc}")

    def status(self):
        """DP_TEMPORAL_* bits of the handle, read without synchronising (include/dragposer.h: dp_temporal_status): 1 once a team of workgroups of
        an earlier launch gave up waiting for a member -- that launch's targets are NaN for the affected sequences, and the next predict() raises"""
        return int(self._lib.dp_temporal_status(self._h))

    def _team_fault(self, team=-1, member=-1, poll_limit=0):
        """tests only: member `member` of sequence `team`'s team never publishes its partial sums; a member gives up after poll_limit re-reads"""
        rc = self._lib.dp_temporal_debug_team_fault(self._h, int(team), int(member), int(poll_limit))
        if rc != 0:
            raise ValueError(f"dp_temporal_debug_team_fault -> {rc}")

    def _team_status(self):
        """tests only (synchronises): 0 = every exchange between the workgroups of a team has completed"""
        return int(self._lib.dp_temporal_debug_team_status(self._h))

    def predict(self, latent_buffer, displacement_buffer, heights_buffer, window, out=None):
        """history buffers [S, H, 24] / [S, H, 3] / [S, H, n_heights] (fp32, on the device, newest entry last)
        -> target_latent_buffer [S, window + 1, 24] (row current_index of it is the frame's z_tgt)"""
        import ctypes as C
        from . import _lib

        S, H = latent_buffer.shape[0], latent_buffer.shape[1]
        for t in (latent_buffer, displacement_buffer, heights_buffer):
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("history buffers must be contiguous fp32 tensors on the predictor's device")
        if out is None:
            out = torch.empty(S, window + 1, 24, device=self.device)
        st = _lib.DpSeqState()
        st.latent_buf, st.disp_buf, st.heights_buf = latent_buffer.data_ptr(), displacement_buffer.data_ptr(), heights_buffer.data_ptr()
        st.history, st.n_heights = H, heights_buffer.shape[2]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.dp_temporal_predict(self._h, S, C.byref(st), int(window), out.data_ptr(), C.c_void_p(stream))
        if rc != _lib.DP_OK:
            msg = self._lib.dp_temporal_last_error(self._h)
            raise _lib.DragPoserError(rc, msg.decode() if msg else "")
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.dp_temporal_destroy(h)
