"""SST proposal encoder (reference: models/sst_model.py:5-40) -- the PRODUCER of `tap_feats`.

SURVEY section 8 marks this "next" (row f-1), outside rows (a)-(e): it is kept on stock PyTorch-ROCm
modules (nn.LSTM -> MIOpen) so that reference-style drivers can construct `models.setup_tap(opt)` and feed
the HIP caption path; it is NOT part of the parity/roofline claims of this round.
"""
import torch
import torch.nn as nn


class SST(nn.Module):
    def __init__(self, opt):
        super(SST, self).__init__()
        self.scores = torch.nn.Linear(opt.hidden_dim, opt.K)
        self.video_dim = opt.video_dim
        self.rnn_type = opt.tap_rnn_type
        self.rnn_num_layers = opt.rnn_num_layers
        self.rnn_dropout = opt.rnn_dropout
        self.K = opt.K
        self.data_for_test = []
        self.rnn = nn.LSTM(opt.video_dim, opt.hidden_dim, opt.rnn_num_layers, batch_first=True, dropout=opt.rnn_dropout)

    def eval(self):                       # the reference only toggles the LSTM's inter-layer dropout (:25-29)
        self.rnn.dropout = 0

    def train(self, mode=True):
        self.rnn.dropout = self.rnn_dropout if mode else 0

    def forward(self, features):
        x = features.unsqueeze(0)                                   # [1,T,D]
        T = x.shape[1]
        h, _ = self.rnn(x)
        h = h.contiguous().view(T, -1)                              # tap_feats [T,hidden]
        return h, torch.sigmoid(self.scores(h)).view(T, self.K)     # proposal scores [T,K]
