"""SST proposal encoder -- the PRODUCER of `tap_feats` for the caption path (reference: models/sst_model.py:5-40).

SURVEY section 8 lists it as a "next" row (f-1), outside rows (a)-(e).  It is kept on stock PyTorch-ROCm modules
(nn.LSTM runs on MIOpen) so that reference-style drivers can build `models.setup_tap(opt)` and feed the HIP caption path;
it takes no part in this round's parity / roofline claims.  Parameter names (`rnn.*`, `scores.*`) match the reference, so
its checkpoints load.
"""
import torch
from torch import nn


class SST(nn.Module):
    """2-layer LSTM over the C3D segment sequence + a K-way sigmoid head (K anchor lengths ending at every segment)."""

    def __init__(self, opt):
        super().__init__()
        self.K = opt.K
        self.video_dim = opt.video_dim
        self.rnn_type = opt.tap_rnn_type
        self.rnn_num_layers = opt.rnn_num_layers
        self.rnn_dropout = opt.rnn_dropout
        self.data_for_test = []
        self.scores = nn.Linear(opt.hidden_dim, self.K)
        self.rnn = nn.LSTM(input_size=opt.video_dim, hidden_size=opt.hidden_dim, num_layers=opt.rnn_num_layers,
                           dropout=opt.rnn_dropout, batch_first=True)

    # The reference overrides train()/eval() so that they ONLY switch the LSTM's inter-layer dropout (sst_model.py:25-29);
    # module.training is left alone on purpose.
    def _set_dropout(self, on):
        self.rnn.dropout = self.rnn_dropout if on else 0

    def train(self, mode=True):
        self._set_dropout(bool(mode))

    def eval(self):
        self._set_dropout(False)

    def forward(self, features):
        """features [T, video_dim] -> (tap_feats [T, hidden_dim], proposal scores [T, K] in (0,1))."""
        T = features.shape[0]
        hidden, _ = self.rnn(features[None])          # one video per call: batch of 1
        tap_feats = hidden.reshape(T, -1)
        return tap_feats, self.scores(tap_feats).sigmoid().reshape(T, self.K)
