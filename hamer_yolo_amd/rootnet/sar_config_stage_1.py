"""rootnet/sar_config_stage_1.py:5-23 (the fields the depth path reads)."""


class rgb_opt:
    backbone = 'resnet34'
    in_channels = 512
    input_img_shape = (256, 256)
    bbox_real = (0.3, 0.3)
    device = 'cuda'
    checkpoint = 'synthetic:0'      # the reference hard-codes /home/pt/fbs/model/rootnet/SAR-resnet34-Root.pth
