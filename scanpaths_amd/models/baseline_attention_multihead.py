"""Drop-in for COCO_Search18/models/baseline_attention_multihead.py:179-424:
baseline(...).forward(images, attention_maps, tasks) with 18 per-category 5x5 head convs (object_sal_layer.<name>)."""
from .scanpath_model import ScanpathModel


class baseline(ScanpathModel):
    def __init__(self, embed_size=512, convLSTM_length=16, min_length=1, ratio=4, map_width=40, map_height=30,
                 arch="resnet50"):
        super().__init__("COCO_Search18", embed_size, convLSTM_length, min_length, ratio, map_width, map_height, arch)

    def forward(self, images, attention_maps, tasks):
        return super().forward(images, attention_maps, tasks)
