"""SST proposal encoder -- the PRODUCER of `tap_feats` for the caption path (reference: models/sst_model.py:5-40;
SURVEY section 8-f row 1).

Parameter names (`rnn.weight_ih_l0` ..., `scores.*`) are nn.LSTM's / nn.Linear's, so the reference's checkpoints load.
On the GPU the arithmetic runs in libechr_hip.so (echr_sst_fwd / echr_sst_bwd: batched input GEMMs + one fused
GEMV+cell launch per timestep); the nn.LSTM module is only the parameter container.  On CPU tensors it falls back to
nothing -- like the rest of echr_amd it raises.
"""
import torch
from torch import nn

from .. import functional as EF


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
        if opt.rnn_num_layers != 2 or str(opt.tap_rnn_type).upper() != 'LSTM':
            raise NotImplementedError('the HIP path implements the shipped SST: a 2-layer LSTM (opts.py:72-78)')
        self.scores = nn.Linear(opt.hidden_dim, self.K)
        self.rnn = nn.LSTM(input_size=opt.video_dim, hidden_size=opt.hidden_dim, num_layers=opt.rnn_num_layers,
                           dropout=opt.rnn_dropout, batch_first=True)
        self._drop_seed = None
        self._drop_calls = 0

    # The reference overrides train()/eval() so that they ONLY switch the LSTM's inter-layer dropout (sst_model.py:25-29);
    # module.training is left alone on purpose.
    def _set_dropout(self, on):
        self.rnn.dropout = self.rnn_dropout if on else 0

    def train(self, mode=True):
        self._set_dropout(bool(mode))

    def eval(self):
        self._set_dropout(False)

    def set_dropout_state(self, seed, calls=0):
        self._drop_seed, self._drop_calls = int(seed), int(calls)

    def native_params(self):
        r = self.rnn
        return (r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0, r.weight_ih_l1, r.weight_hh_l1, r.bias_ih_l1, r.bias_hh_l1,
                self.scores.weight, self.scores.bias)

    def forward(self, features):
        """features [T, video_dim] -> (tap_feats [T, hidden_dim], proposal scores [T, K] in (0,1))."""
        if not features.is_cuda:
            raise EF.L.EchrHipError('SST runs on the GPU only: move the module and its inputs with .cuda()')
        p = float(self.rnn.dropout)
        if self._drop_seed is None:
            self._drop_seed = (int(torch.initial_seed()) ^ 0x55AA) & 0xFFFFFFFFFFFFFFFF
        drop = EF.DropState(self._drop_seed, self._drop_calls, p > 0.0)
        if p > 0.0:
            self._drop_calls += 1
        arena = getattr(self, '_echr_arena', None)
        sink = EF.GradSink(arena, self.native_params()) if arena is not None else None
        return EF.SSTFunction.apply(features, p, drop, sink, *self.native_params())

    def build_arena(self):
        """Pack parameters and gradients into flat device buffers (echr_amd/arena.py): ClampAdam(..., arena=...) then updates the whole
        proposal encoder in ONE launch, as CaptionGenerator.build_arena() does for the caption path.  Call after .cuda()."""
        from ..arena import ParamArena
        return ParamArena(self)
