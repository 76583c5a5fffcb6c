"""IR-SE units of the ReStyle/pSp encoder with ``ModuleList`` bodies (reference backbone/restyle_psp_helpers.py).

Stage 2 (restyle-encoder) stores these blocks as ``Sequential``; Stage 3 re-creates them as ``ModuleList`` with the
same child indices so ``res_layer.{0..5}`` / ``shortcut_layer.{0,1}`` checkpoint keys line up, and so Dropout can
be spliced in later (``add_dropout``, reference :201-209).  As in backbone/model_irse.py the classes only own
parameters; execution is the frhip engine's.  The experimental demographic-adaptive layers of the reference file
(Conv2dExtended, AdaConv2d_faster, AttBlock) are never instantiated by ``pSp`` and are out of scope.
"""
from torch.nn import BatchNorm2d, Conv2d, Dropout, MaxPool2d, Module, ModuleList, PReLU

from backbone.model_irse import SEModule, _conv3x3, _eager_forbidden, l2_norm  # noqa: F401

_STAGES = {34: (3, 4, 6, 3), 50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}
_WIDTHS = ((64, 64), (64, 128), (128, 256), (256, 512))


def get_blocks(num_layers):
    if num_layers not in _STAGES:
        raise ValueError("Invalid number of layers: {}. Must be one of [34, 50, 100, 152]".format(num_layers))
    return [[(cin, depth, 2)] + [(depth, depth, 1)] * (n - 1) for (cin, depth), n in zip(_WIDTHS, _STAGES[num_layers])]


class bottleneck_IR_SE(Module):
    def __init__(self, in_channel, depth, stride, dropout=None):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = ModuleList([Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth)])
        self.res_layer = ModuleList([BatchNorm2d(in_channel), _conv3x3(in_channel, depth, 1), PReLU(depth),
                                     _conv3x3(depth, depth, stride), BatchNorm2d(depth), SEModule(depth, 16)])
        self.in_channel, self.depth, self.use_att = in_channel, depth, False
        if dropout:
            print("[bottleneck_IR_SE] adding dropout layer")
            self.add_dropout(dropout)

    def forward(self, x, race=None):
        """A unit called on its own (reference :190-199; ``race`` feeds the experimental adaptive layers only)."""
        from frhip.engine import run_unit
        return run_unit(self, x)

    def add_dropout(self, p):
        """Dropout after the shortcut conv and after each residual conv (list indices 1 / 2 and 5)."""
        if isinstance(self.shortcut_layer, ModuleList):
            self.shortcut_layer.insert(1, Dropout(p=p))
        self.res_layer.insert(2, Dropout(p=p))
        self.res_layer.insert(5, Dropout(p=p))
