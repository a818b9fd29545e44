"""CPU: the temporal predictor module and DragPose's temporal target block against what the
reference's own Temporal / DragPose.run produced (tests/golden/seq*.npz, frame 0 of every sequence:
the block only needs the initial state there, no kernel)."""
import os

import numpy as np
import pytest
import torch

from dragposer_amd.temporal import HISTORY, TemporalPredictor
from oracle import ref_torch as R


def _load_temporal(g):
    tp = g["meta"]["temporal_param"]
    m = TemporalPredictor(n_encoder_layers=tp["n_encoder_layers"], n_decoder_layers=tp["n_decoder_layers"],
                          dim_feedforward=tp["dim_feedforward"])
    sd = {k[len("temporal."):]: torch.tensor(v) for k, v in g.items() if k.startswith("temporal.")}
    missing = m.load_state_dict(sd, strict=True)
    return m.eval()


def test_full_size_architecture_matches_reference_parameter_count():
    n = sum(p.numel() for p in TemporalPredictor().parameters())
    assert n == 1282536, n  # "# parameters temporal: 1282536" is what the reference prints for its shipped hyper-parameters


@pytest.mark.parametrize("name", ["seq6", "seq3", "seq4"])
def test_temporal_block_first_frame(golden_dir, name):
    from dragposer_amd.drag_pose import DragPose

    g = R.load_golden(os.path.join(golden_dir, f"{name}.npz"))
    K = g["meta"]["K"]
    window = g["meta"]["cfg"]["temporal_future_window"]

    class _NoKernel:  # DragPose's temporal block needs only the device and the model statistics
        device = torch.device("cpu")

        class host_model:
            arrays = dict(mean_q=np.zeros(88, np.float32), std_q=np.ones(88, np.float32), offsets=np.zeros((22, 3), np.float32))

    dp = DragPose(_NoKernel(), _load_temporal(g), g["means_latent"], g["stds_latent"], n_sequences=K)
    dp.set_initial_state(g["z0"], np.zeros((K, 3), np.float32), g["init_rot"], g["init_heights"])
    assert dp.latent_buffer.shape == (K, HISTORY, 24)
    dp._temporal_targets(window)
    np.testing.assert_allclose(dp.target_latent_buffer[:, 0].numpy(), g["z_tgt"][0], atol=2e-6)
