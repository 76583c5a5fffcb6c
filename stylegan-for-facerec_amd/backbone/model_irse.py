"""IR / IR-SE face-recognition backbones with the reference's import path and state-dict layout.

    from backbone.model_irse import IR_50, IR_101, IR_152, IR_SE_50, IR_SE_101, IR_SE_152   (reference train.py:7)

The module tree below exists to *own parameters under the reference's names* (``input_layer.{0,1,2}``,
``body.{i}.res_layer.{0..5}``, ``body.{i}.shortcut_layer.{0,1}``, ``output_layer.{0,3,4}`` -- SURVEY.md App. B
item 12; reference backbone/model_irse.py:129-163) so checkpoints, ``separate_irse_bn_paras`` and the freeze
logic of the training loop keep working.  It does not compute: ``Backbone.forward`` hands the batch to
``frhip.engine`` which executes the whole network as a static list of fused HIP launches (NHWC activations,
MFMA implicit-GEMM convolutions with BatchNorm/PReLU folded into their gathers and epilogues).  Leaves are
stock ``torch.nn`` layers on purpose: the weight-decay split is decided by their class names.

Leaf layers raise if they are ever called directly, and the residual units run through the engine on their own
(``frhip.engine.UnitStackRunner``): there is no CPU / eager fallback in the product path (the CPU restatement used for
parity lives in ``oracle/`` and is test-only).
"""
import torch
import torch.nn as nn
from torch.nn import BatchNorm1d, BatchNorm2d, Conv2d, Dropout, Linear, MaxPool2d, Module, PReLU, Sequential

from frhip.engine import BackboneRunner, run_unit

# stage plan: (width in, width out) and the number of units per stage for each depth
_STAGE_WIDTHS = ((64, 64), (64, 128), (128, 256), (256, 512))
_STAGE_UNITS = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}


def _eager_forbidden(self, *args, **kwargs):
    raise RuntimeError("%s is a parameter holder of the frhip HIP engine and is not executed eagerly; call the "
                       "enclosing backbone instead" % type(self).__name__)


class Flatten(Module):
    forward = _eager_forbidden


def l2_norm(input, axis=1):
    return input / torch.norm(input, 2, axis, True)


def _conv3x3(cin, cout, stride):
    conv = Conv2d(cin, cout, (3, 3), stride, 1, bias=False)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)  # packed [O][kh][kw][I]
    return conv


class SEModule(Module):
    """Squeeze-and-excite parameters: fc1 [C/r, C, 1, 1], fc2 [C, C/r, 1, 1], no biases."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        nn.init.xavier_uniform_(self.fc1.weight.data)
        self.relu = nn.ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = nn.Sigmoid()

    forward = _eager_forbidden


class bottleneck_IR(Module):
    """One pre-activation residual unit: BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN, plus shortcut."""
    _with_se = False

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)  # pure strided subsampling
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))
        layers = [BatchNorm2d(in_channel), _conv3x3(in_channel, depth, 1), PReLU(depth),
                  _conv3x3(depth, depth, stride), BatchNorm2d(depth)]
        if self._with_se:
            layers.append(SEModule(depth, 16))
        self.res_layer = Sequential(*layers)

    def forward(self, x):
        """A unit called on its own (inside a Backbone the engine runs the whole network instead): the HIP launch lists
        of this one unit, with gradients for x and the parameters."""
        return run_unit(self, x)


class bottleneck_IR_SE(bottleneck_IR):
    _with_se = True


def get_blocks(num_layers):
    """[[(in_channel, depth, stride), ...] per stage]; the first unit of every stage has stride 2."""
    return [[(cin, depth, 2)] + [(depth, depth, 1)] * (n - 1)
            for (cin, depth), n in zip(_STAGE_WIDTHS, _STAGE_UNITS[num_layers])]


class Backbone(Module):
    def __init__(self, input_size, num_layers, mode="ir"):
        super().__init__()
        assert input_size[0] in [112, 224], "input_size should be [112, 112] or [224, 224]"
        assert num_layers in [50, 100, 152], "num_layers should be 50, 100 or 152"
        assert mode in ["ir", "ir_se"], "mode should be ir or ir_se"
        self.input_size = list(input_size)
        unit = bottleneck_IR if mode == "ir" else bottleneck_IR_SE
        side = input_size[0] // 16
        # registration order input_layer, output_layer, body matches the reference's state-dict order
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        self.output_layer = Sequential(BatchNorm2d(512), Dropout(), Flatten(), Linear(512 * side * side, 512),
                                       BatchNorm1d(512))
        self.body = Sequential(*[unit(*spec) for stage in get_blocks(num_layers) for spec in stage])
        self._initialize_weights()
        self._runner = [BackboneRunner(self, in_channels=3)]  # list: keep it out of the module registry

    def forward(self, x):
        return self._runner[0](x)

    def _initialize_weights(self):
        """xavier-uniform convs / linears, zero biases, BN (1, 0) -- reference model_irse.py:174-189."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = [BackboneRunner(new, in_channels=3)] if k == "_runner" else copy.deepcopy(v, memo)
        return new


def IR_50(input_size):
    return Backbone(input_size, 50, "ir")


def IR_101(input_size):
    return Backbone(input_size, 100, "ir")


def IR_152(input_size):
    return Backbone(input_size, 152, "ir")


def IR_SE_50(input_size):
    return Backbone(input_size, 50, "ir_se")


def IR_SE_101(input_size):
    return Backbone(input_size, 100, "ir_se")


def IR_SE_152(input_size):
    return Backbone(input_size, 152, "ir_se")
