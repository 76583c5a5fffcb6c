"""``from backbone.model_resnet import ResNet_50, ResNet_101, ResNet_152`` (reference train.py:6).

These torchvision-style ResNets are imported by the reference driver but are not named by any shipped config nor
by BASELINE.json; they are outside the accelerated path and deliberately not built: selecting one
fails loudly instead of silently running an un-accelerated eager model.
"""


def _off_path(name):
    def ctor(input_size):
        raise NotImplementedError("%s is outside the scope of the frhip build (SURVEY.md section 2.1): only the "
                                  "IR / IR-SE / IR_*_ReStyle backbones are implemented" % name)
    ctor.__name__ = name
    return ctor


ResNet_50 = _off_path("ResNet_50")
ResNet_101 = _off_path("ResNet_101")
ResNet_152 = _off_path("ResNet_152")
