"""torchvision ResNet-34 as the reference wraps it (``SARresnet34``, rootnet/Model_RGB.py:179-196):
``extract_mid = Sequential(conv1, bn1, relu, maxpool, layer1, layer2)``, ``extract_high = ModuleList([Sequential(layer3,
layer4)])`` -- hence the state-dict prefixes below.  BasicBlock: conv3x3(s)-bn-relu-conv3x3-bn, ``+= identity`` (or
``downsample = conv1x1(s)-bn`` of the input), relu.  torchvision itself is not installed here; the layout is the
published one (layers [3, 4, 6, 3], widths [64, 128, 256, 512], stem 7x7/2 + maxpool 3x3/2)."""
from typing import List, Tuple

LAYERS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]           # (width, blocks, stride of the first block)
LAYER_PREFIX = ["backbone.extract_mid.4.", "backbone.extract_mid.5.", "backbone.extract_high.0.0.", "backbone.extract_high.0.1."]
STEM_CONV, STEM_BN = "backbone.extract_mid.0", "backbone.extract_mid.1"
BN_EPS = 1e-5


def blocks() -> List[Tuple[str, int, int, int, bool]]:
    """(prefix, cin, cout, stride, has_downsample) for the 16 BasicBlocks in forward order."""
    out, cin = [], 64
    for (width, n, stride), pre in zip(LAYERS, LAYER_PREFIX):
        for i in range(n):
            s = stride if i == 0 else 1
            out.append((f"{pre}{i}.", cin, width, s, s != 1 or cin != width))
            cin = width
    return out


def conv_specs():
    """name -> (cin, cout, k, stride) of every convolution (BN folded in), stem first."""
    specs = {"stem": (3, 64, 7, 2)}
    for pre, cin, cout, s, ds in blocks():
        specs[pre + "conv1"] = (cin, cout, 3, s)
        specs[pre + "conv2"] = (cout, cout, 3, 1)
        if ds:
            specs[pre + "downsample"] = (cin, cout, 1, s)
    return specs
