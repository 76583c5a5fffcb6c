"""``from backbone.restyle_psp import pSp`` -- the Stage-2 ReStyle/pSp encoder trunk reused as FR backbone.

Reference: backbone/restyle_psp.py:358-478 (``pSp``) and :118-216 (``BackboneEncoderDiffHead``).  ``pSp`` wraps an
IR-SE encoder whose stem takes **6 channels**: the input image concatenated with a constant average image
(restyle_psp.py:445-447); the output head is the face-recognition one (BN -> Dropout -> Flatten -> Linear -> BN1d).
Only ``encoder.input_layer.*`` and ``encoder.body.*`` are taken from a Stage-2 checkpoint (:419-437).

Execution: the frhip engine.  The concatenation is never materialised -- the stem's im2col kernel reads the
three image channels from the batch and the other three from ``avg_image`` directly.

Observable differences from the reference: ``avg_image`` may be a path (read with PIL instead of imageio) *or* a
``[3,H,W]`` tensor; nothing is pinned to ``'cuda:0'`` (reference :385,389) -- buffers follow the module's device.
"""
import torch
import torch.nn as nn
from torch.nn import BatchNorm2d, Conv2d, Module, PReLU, Sequential

from backbone.restyle_psp_helpers import bottleneck_IR_SE, get_blocks
from frhip.engine import BackboneRunner
from util.utils import _initialize_weights

_END_SIZE = {400: 25, 256: 16, 200: 13, 112: 7}


class BackboneEncoderDiffHead(Module):
    def __init__(self, num_layers, mode="ir", n_styles=18, opts=None, emb_size=512, input_size=256,
                 double_in_channels=False, output_layer_type="facerec", include_dropout=None):
        super().__init__()
        assert num_layers in [34, 50, 100, 152], "num_layers should be 34, 50,100, or 152"
        assert mode in ["ir", "ir_se"], "mode should be ir or ir_se"
        if mode != "ir_se":
            raise TypeError("bottleneck_IR.__init__() got an unexpected keyword argument 'dropout'")
        if output_layer_type != "facerec":
            raise NotImplementedError("only the 'facerec' output head is on the Stage-3 path")
        print("Initializing backbone encoder with {} layers".format(num_layers))
        self.input_size = input_size
        self.input_layer = Sequential(Conv2d(6, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        self.input_layer_att = nn.ModuleList([])
        k = 2 if double_in_channels else 1
        self.body = nn.ModuleList([bottleneck_IR_SE(cin * k, depth, stride, dropout=include_dropout)
                                   for stage in get_blocks(num_layers) for cin, depth, stride in stage])
        side = _END_SIZE[input_size]
        self.output_layer_type = output_layer_type
        self.output_layer = Sequential(nn.BatchNorm2d(512), nn.Dropout(), nn.Flatten(),
                                       nn.Linear(512 * side * side, emb_size), nn.BatchNorm1d(emb_size))
        self.use_att = False
        self._runner = [BackboneRunner(self, in_channels=6)]

    def add_dropouts(self, include_dropout=None):
        if include_dropout:
            for name, m in self.body.named_modules():
                if isinstance(m, bottleneck_IR_SE):
                    print("adding dropout to the layer", name)
                    m.add_dropout(include_dropout)

    def forward(self, x, *args, races=None, avg_image=None, **kwargs):
        if x.shape[1] == 6:  # caller concatenated already, as the reference does
            x, avg_image = x[:, :3].contiguous(), x[0, 3:].contiguous()
        return self._runner[0](x, avg_image)


class pSp(nn.Module):
    def __init__(self, size=256, encoder_type="BackboneEncoder", checkpoint_path=None, avg_image=None,
                 num_diff_blocks=1, include_dropout=None, include_attblocks=None, attblock_init_strategy="ones",
                 decoder_checkpoint_path=None, subbatch_mode="random", stylegan_subbatch_size=None):
        super().__init__()
        self.size = size
        self.encoder_type = encoder_type
        self.num_diff_blocks = num_diff_blocks
        self.include_dropout = include_dropout
        self.subbatch_mode = subbatch_mode
        self.stylegan_subbatch_size = stylegan_subbatch_size
        self.encoder = self.set_encoder(encoder_type)
        self.avg_image = None if avg_image is None else self._load_avg_image(avg_image)
        _initialize_weights(self.encoder)
        if checkpoint_path is not None:
            print("[pSp] Loading weights...")
            self.load_weights(checkpoint_path)
        if include_dropout:
            print("[include_dropout]")
            self.encoder.add_dropouts(include_dropout)

    @staticmethod
    def _load_avg_image(avg_image):
        """uint8 HWC image (or path) -> float CHW in [-1, 1] (reference :381-389)."""
        if isinstance(avg_image, torch.Tensor):
            return avg_image.detach().float()
        import numpy as np
        from PIL import Image
        if str(avg_image).endswith(".npy"):  # uint8 HWC array (what imageio.imread returns in the reference)
            arr = np.load(avg_image)
        else:
            arr = np.asarray(Image.open(avg_image).convert("RGB"))
        t = torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255.0
        return ((t - 0.5) / 0.5).detach()

    def set_encoder(self, encoder_type):
        layers = {"BackboneEncoder": 50, "BackboneEncoder34": 34, "BackboneEncoder100": 100}.get(encoder_type)
        if layers is None:
            raise Exception(f"{encoder_type} is not a valid encoders")
        return BackboneEncoderDiffHead(layers, "ir_se", input_size=self.size)

    def load_weights(self, checkpoint_path):
        if checkpoint_path is None:
            print("NOT loading weights for encoder (checkpoint_path is None)")
            return
        print(f"Loading ReStyle pSp from checkpoint: {checkpoint_path}")
        ckpt = torch.load(checkpoint_path, map_location="cpu")
        enc = self._sub_keys(ckpt, "encoder")
        print("Loading encoder weights...")
        self.encoder.input_layer.load_state_dict(self._sub_keys(enc, "input_layer"), strict=True)
        self.encoder.body.load_state_dict(self._sub_keys(enc, "body"), strict=True)

    @staticmethod
    def _sub_keys(d, name):
        """Entries of a (possibly ``{'state_dict': ...}``-wrapped) dict whose key starts with ``name``, prefix and
        the following separator removed."""
        d = d["state_dict"] if "state_dict" in d else d
        cut = len(name) + 1
        return {k[cut:]: v for k, v in d.items() if k[:len(name)] == name}

    def forward(self, x, races=None, is_generated=None):
        if x.size(2) != self.size:
            # F.interpolate(x, self.size, mode='bilinear') of the reference (:440-443), one HIP launch
            print('[interpolating ', x.size(2), ' to ', self.size, ']')
            from frhip import _lib, ops
            if not x.is_cuda:
                raise _lib.FrhipError("frhip: pSp runs on the HIP path only -- got a %s tensor" % x.device)
            xin = x.contiguous().float()
            x = torch.empty(xin.shape[0], xin.shape[1], self.size, self.size, device=xin.device)
            ops.call("fr_resize_bilinear", xin, x, xin.shape[0] * xin.shape[1], xin.shape[2], xin.shape[3], self.size,
                     self.size, ops.current_stream_ptr())()
        return self.encoder(x, races=races, avg_image=self.avg_image)
