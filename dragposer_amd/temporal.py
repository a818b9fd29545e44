"""The temporal predictor, kept in PyTorch (ROCm) as the reference keeps it.

`TemporalPredictor` has the reference's architecture and state_dict layout
(python/src/temporal_transformer.py:7-77, positional_encoding.py:6-32; hyper-parameters
train_temporal.py:17-37), so a ``temporal.pt`` trained with the reference loads unchanged via
`load_reference_checkpoint`.  It produces `target_latent`, the anchor of the lambda_temporal pull
term; it is not part of the HIP kernel.
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
    model = TemporalPredictor(**arch).to(device)
    model.load_state_dict(ck["model_state_dict"])
    return model.eval(), ck["means_latent"].to(device), ck["stds_latent"].to(device)
