"""The pose-VAE encoder (reference autoencoder.py:56-143, skeleton.py:8-130,178-210), used once per
sequence to get the initial latent (drag_pose.py:47-51).  Plain PyTorch on the optimiser's device: three
[masked dense (kernel-size-1 SkeletonConv) -> SkeletonPool -> LeakyReLU(0.2)] stages 176->112->72->48, then
f_mu / f_logvar.  Not part of the per-frame hot path."""
import numpy as np
import torch

from .model import DEFAULT_MODEL


class PoseEncoder(torch.nn.Module):
    def __init__(self, model_path=DEFAULT_MODEL, arrays=None):
        super().__init__()
        raw = arrays if arrays is not None else np.load(model_path)
        t = lambda k: torch.tensor(np.asarray(raw[k]), dtype=torch.float32)
        for l in range(3):
            w = t(f"encoder.layers.{l}.0.weight")[..., 0] * t(f"encoder.layers.{l}.0.mask")[..., 0]  # skeleton.py:120
            self.register_buffer(f"conv_w{l}", w)
            self.register_buffer(f"conv_b{l}", t(f"encoder.layers.{l}.0.bias"))
            self.register_buffer(f"pool_w{l}", t(f"encoder.layers.{l}.1.weight"))
        self.register_buffer("mu_w", t("encoder.f_mu.weight"))
        self.register_buffer("mu_b", t("encoder.f_mu.bias"))
        self.register_buffer("lv_w", t("encoder.f_logvar.weight"))
        self.register_buffer("lv_b", t("encoder.f_logvar.bias"))

    def forward(self, pose):
        """pose [S, 176] normalised dual quaternions -> mu [S, 24], logvar [S, 24]"""
        h = pose
        for l in range(3):
            h = h @ getattr(self, f"conv_w{l}").T + getattr(self, f"conv_b{l}")
            h = h @ getattr(self, f"pool_w{l}").T
            h = torch.nn.functional.leaky_relu(h, 0.2)
        return h @ self.mu_w.T + self.mu_b, h @ self.lv_w.T + self.lv_b

    def sample(self, pose, generator=None, use_mean=False):
        """latent as the reference draws it: mu + eps * exp(0.5 logvar) (autoencoder.py:19-27)"""
        mu, logvar = self.forward(pose)
        if use_mean:
            return mu
        eps = torch.randn(mu.shape, generator=generator, device="cpu").to(mu.device)
        return mu + eps * torch.exp(0.5 * logvar)
